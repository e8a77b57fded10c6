// pdbatch host side: track loader.  Reads the reference's on-disk track formats and emits the
// packed blob the kernels consume.
//   surfaces.bin  Sim/Track.cpp:97-150 (58-byte packed BlobSurface, Sim/Surface.h:25-45)
//   spline.bin    Sim/Track.cpp:274-293 (SlimTrackPoint, Sim/Track.h:12-17)
//   spline.cache  Sim/Track.cpp:294-311 (FatTrackPoint, Sim/Track.h:18-25)
//   spline.ini    Sim/Track.cpp:158-186
//   pits.ini      Sim/Track.cpp:151-175 (Track::loadPits)
// and restates the load-time geometry: computeFatPoints / computeSideLocation (Track.cpp:366-467, both the
// TRACE_SIDES=0 and the ray-traced sides; pinned by the spline.cache files the reference ships), initTrackPoints (Track.cpp:188-272) and BSpline3d::init_from_array (Core/Spline3d.cpp:79-162).
#include "model.hpp"
#include "ini.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>

namespace pdb {

#pragma pack(push, 1)
struct BlobSurface {
    uint32_t magic, numVertices, numIndices, sectorID, collisionCategory;
    float gripMod, damping, sinHeight, sinLength, granularity, dirtAdditiveK, vibrationGain, vibrationLength, wavPitchSpeed;
    uint8_t isValidTrack, isPitlane;
};
#pragma pack(pop)
static_assert(sizeof(BlobSurface) == 58, "BlobSurface layout");

// One vertical-or-general ray against the triangle soup.  Moeller-Trumbore in OPCODE's culling form
// (det = e1 . (dir x e2) must exceed 1e-6); per surface mesh the nearest hit, across meshes a
// strictly nearer hit replaces the current one (PhysicsEngineODE.cpp:196-211); normal =
// normalise((v1-v0) x (v2-v0)).
static inline bool rayTri(const float* o, const float* d, float maxDist, const float* v0, const float* v1, const float* v2, float& tOut) {
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det < 1.0e-6f) return false;
    const float tv[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
    const float u = tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2];
    if (u < 0.0f || u > det) return false;
    const float q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float v = d[0] * q[0] + d[1] * q[1] + d[2] * q[2];
    if (v < 0.0f || u + v > det) return false;
    float t = e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2];
    t *= 1.0f / det;
    if (t < 0.0f || !(t < maxDist)) return false;
    tOut = t;
    return true;
}

static RayHitH rayCastRaw(const pdb_surface* surfaces, int numSurfaces, const float* tris, const float* o, const float* d, float maxDist) {
    RayHitH best;
    for (int s = 0; s < numSurfaces; ++s) {
        float bt = -1.0f; int btri = -1;
        for (int t = surfaces[s].triStart; t < surfaces[s].triStart + surfaces[s].triCount; ++t) {
            float tt;
            if (rayTri(o, d, maxDist, tris + 9 * t, tris + 9 * t + 3, tris + 9 * t + 6, tt))
                if (bt < 0.0f || tt < bt) { bt = tt; btri = t; }
        }
        if (btri >= 0 && (best.depth < 0.0f || best.depth > bt)) {
            const float* v0 = tris + 9 * btri; const float* v1 = v0 + 3; const float* v2 = v0 + 6;
            const float vu[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
            const float vv[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
            float n[3] = {vu[1] * vv[2] - vu[2] * vv[1], vu[2] * vv[0] - vu[0] * vv[2], vu[0] * vv[1] - vu[1] * vv[0]};
            const float l = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
            if (l > 0.0f) {
                const float sc = 1.0f / sqrtf(l);
                best.has = true; best.depth = bt; best.surface = s;
                for (int k = 0; k < 3; ++k) { best.pos[k] = o[k] + d[k] * bt; best.normal[k] = n[k] * sc; }
            }
        }
    }
    return best;
}

RayHitH rayCastTrack(const TrackView& tv, const float* origin, const float* dir, float maxDist) {
    return rayCastRaw(tv.surfaces, tv.h->numSurfaces, tv.tris, origin, dir, maxDist);
}

static std::vector<uint8_t> readFile(const std::string& p) {
    std::vector<uint8_t> d;
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return d;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    d.resize((size_t)n);
    if (n > 0 && fread(d.data(), 1, (size_t)n, f) != (size_t)n) d.clear();
    fclose(f);
    return d;
}

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, float f) { return {a.x * f, a.y * f, a.z * f}; }
static inline V3 operator*(float f, V3 a) { return {a.x * f, a.y * f, a.z * f}; }
static inline V3 operator/(V3 a, float f) { return {a.x / f, a.y / f, a.z / f}; }
static inline float len(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline V3 norm(V3 a) { const float l = len(a); if (l != 0.0f) return a * (1.0f / l); return a; }
static inline V3 cross(V3 a, V3 v) { return {a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x}; }

// Core/Spline3d.cpp:154-162
static V3 bsInterp(float u, V3 P0, V3 P1, V3 P2, V3 P3) {
    V3 p = u * u * u * ((-1.0f) * P0 + 3.0f * P1 - 3.0f * P2 + P3) / 6.0f;
    p = p + u * u * (3.0f * P0 - 6.0f * P1 + 3.0f * P2) / 6.0f;
    p = p + u * (-3.0f * P0 + 3.0f * P2) / 6.0f;
    p = p + (P0 + 4.0f * P1 + P2) / 6.0f;
    return p;
}

// Rays of one side trace share their origin and lie in one vertical plane, so their xz footprint is a segment: the
// triangles whose xz box meets that segment (slab test) are a superset of everything any of the rays can hit.  The ids
// stay in ascending order, so the per-mesh "nearest, first on ties" rule sees the candidates in the order of a full scan.
static void trisNearSegment(const std::vector<float>& tris, float ax, float az, float bx, float bz, std::vector<int32_t>& out) {
    out.clear();
    const float dx = bx - ax, dz = bz - az;
    const float eps = 1.0e-3f;
    const size_t nt = tris.size() / 9;
    for (size_t t = 0; t < nt; ++t) {
        const float* p = tris.data() + 9 * t;
        const float x0 = std::min(p[0], std::min(p[3], p[6])) - eps, x1 = std::max(p[0], std::max(p[3], p[6])) + eps;
        const float z0 = std::min(p[2], std::min(p[5], p[8])) - eps, z1 = std::max(p[2], std::max(p[5], p[8])) + eps;
        float t0 = 0.0f, t1 = 1.0f;
        bool ok = true;
        if (dx != 0.0f) { float u0 = (x0 - ax) / dx, u1 = (x1 - ax) / dx; if (u0 > u1) std::swap(u0, u1); t0 = std::max(t0, u0); t1 = std::min(t1, u1); }
        else if (ax < x0 || ax > x1) ok = false;
        if (dz != 0.0f) { float u0 = (z0 - az) / dz, u1 = (z1 - az) / dz; if (u0 > u1) std::swap(u0, u1); t0 = std::max(t0, u0); t1 = std::min(t1, u1); }
        else if (az < z0 || az > z1) ok = false;
        if (ok && t0 <= t1) out.push_back((int32_t)t);
    }
}

// rayCastRaw restricted to a candidate list (ascending triangle ids; triSurf maps triangle -> surface)
static RayHitH rayCastSubset(const std::vector<int32_t>& cand, const std::vector<int32_t>& triSurf, const float* tris, const float* o, const float* d, float maxDist) {
    RayHitH best;
    size_t i = 0;
    while (i < cand.size()) {
        const int32_t s = triSurf[(size_t)cand[i]];
        float bt = -1.0f; int btri = -1;
        for (; i < cand.size() && triSurf[(size_t)cand[i]] == s; ++i) {
            const int t = cand[i];
            float tt;
            if (rayTri(o, d, maxDist, tris + 9 * t, tris + 9 * t + 3, tris + 9 * t + 6, tt))
                if (bt < 0.0f || tt < bt) { bt = tt; btri = t; }
        }
        if (btri >= 0 && (best.depth < 0.0f || best.depth > bt)) {
            const float* v0 = tris + 9 * btri; const float* v1 = v0 + 3; const float* v2 = v0 + 6;
            const float vu[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
            const float vv[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
            const float n[3] = {vu[1] * vv[2] - vu[2] * vv[1], vu[2] * vv[0] - vu[0] * vv[2], vu[0] * vv[1] - vu[1] * vv[0]};
            const float l = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
            if (l > 0.0f) {
                const float sc = 1.0f / sqrtf(l);
                best.has = true; best.depth = bt; best.surface = s;
                for (int k = 0; k < 3; ++k) { best.pos[k] = o[k] + d[k] * bt; best.normal[k] = n[k] * sc; }
            }
        }
    }
    return best;
}

std::vector<uint8_t> buildTrack(const std::string& basePathIn, const std::string& trackName, bool recomputeFatPoints) {
    std::string base = basePathIn;
    std::replace(base.begin(), base.end(), '\\', '/');
    if (!base.empty() && base.back() != '/') base += '/';
    const std::string dir = base + "content/tracks/" + trackName + "/";

    float grip = 1.0f, cellSize = 50.0f;
    Ini simIni(base + "cfg/sim.ini");
    if (simIni.ready) { simIni.tryGetFloat("ENVIRONMENT", "TRACK_GRIP", grip); simIni.tryGetFloat("VERTEX_HASH", "CELL_SIZE", cellSize); }

    // ---- surfaces ----
    std::vector<pdb_surface> surfaces;
    std::vector<float> tris;
    {
        const std::vector<uint8_t> d = readFile(dir + "surfaces.bin");
        if (d.empty()) throw std::runtime_error("pdb: cannot read " + dir + "surfaces.bin");
        size_t pos = 0;
        while (pos + sizeof(BlobSurface) <= d.size()) {
            BlobSurface b;
            memcpy(&b, d.data() + pos, sizeof(b));
            pos += sizeof(b);
            if (b.magic != 0xAABBCCDD || !b.numVertices || !b.numIndices) throw std::runtime_error("pdb: bad surface blob");
            const size_t vb = (size_t)b.numVertices * 12, ib = (size_t)b.numIndices * 2;
            if (pos + vb + ib > d.size()) throw std::runtime_error("pdb: truncated surface blob");
            const float* v = reinterpret_cast<const float*>(d.data() + pos);
            const uint16_t* ix = reinterpret_cast<const uint16_t*>(d.data() + pos + vb);
            pdb_surface s;
            memset(&s, 0, sizeof(s));
            s.gripMod = b.gripMod; s.damping = b.damping; s.sinHeight = b.sinHeight; s.sinLength = b.sinLength;
            s.granularity = b.granularity; s.dirtAdditiveK = b.dirtAdditiveK;
            s.collisionCategory = (int32_t)b.collisionCategory; s.isValidTrack = b.isValidTrack; s.sectorID = (int32_t)b.sectorID;
            if (b.collisionCategory > 127u) throw std::runtime_error("pdb: surfaces.bin collision category out of range (the C_CATEGORY_* bits are 1..16)");
            s.triStart = (int32_t)(tris.size() / 9);
            s.triCount = (int32_t)(b.numIndices / 3);
            for (uint32_t t = 0; t < b.numIndices / 3; ++t)
                for (int k = 0; k < 3; ++k) {
                    float vv[3];
                    if (ix[3 * t + k] >= b.numVertices) throw std::runtime_error("pdb: surface index out of range in " + trackName + "/surfaces.bin");
                    memcpy(vv, reinterpret_cast<const uint8_t*>(v) + (size_t)ix[3 * t + k] * 12, 12);
                    tris.push_back(vv[0]); tris.push_back(vv[1]); tris.push_back(vv[2]);
                }
            surfaces.push_back(s);
            pos += vb + ib;
        }
    }

    // ---- spline ----
    bool closedLoop = false, traceSides = false;
    float traceRayOffsetY = 20.0f, traceRayLength = 100.0f, traceSideMax = 10.0f, traceDiffHeightMax = 0.01f, traceDiffGripMax = 0.1f, traceStep = 0.01f;   // Sim/Track.h:83-89
    std::vector<int> traceBadSectors;
    Ini spl(dir + "spline.ini");
    if (spl.ready) {
        closedLoop = spl.getInt("SPLINE", "CLOSED_LOOP") != 0;
        traceSides = spl.getInt("SPLINE", "TRACE_SIDES") != 0;
        spl.tryGetFloat("SPLINE", "TRACE_RAY_OFFSET_Y", traceRayOffsetY);
        spl.tryGetFloat("SPLINE", "TRACE_RAY_LENGTH", traceRayLength);
        spl.tryGetFloat("SPLINE", "TRACE_SIDE_MAX", traceSideMax);
        spl.tryGetFloat("SPLINE", "TRACE_DIFF_HEIGHT_MAX", traceDiffHeightMax);
        spl.tryGetFloat("SPLINE", "TRACE_DIFF_GRIP_MAX", traceDiffGripMax);
        spl.tryGetFloat("SPLINE", "TRACE_STEP", traceStep);
        if (spl.hasKey("SPLINE", "TRACE_BAD_SECTORS")) {   // "2|4" (Track.cpp:194-203)
            const std::string list = spl.getString("SPLINE", "TRACE_BAD_SECTORS");
            size_t a = 0;
            while (a <= list.size()) {
                size_t b = list.find('|', a);
                if (b == std::string::npos) b = list.size();
                if (b > a) traceBadSectors.push_back(atoi(list.substr(a, b - a).c_str()));
                a = b + 1;
            }
        }
    }
    struct Slim { float best[3]; float sides[2]; };
    struct Fat { V3 best, left, right, center, forwardDir; };
    std::vector<Slim> slim;
    {
        const std::vector<uint8_t> d = readFile(dir + "spline.bin");
        slim.resize(d.size() / sizeof(Slim));
        if (!slim.empty()) memcpy(slim.data(), d.data(), slim.size() * sizeof(Slim));
    }
    std::vector<Fat> fat;
    {
        const std::vector<uint8_t> d = readFile(dir + "spline.cache");
        fat.resize(d.size() / sizeof(Fat));
        if (!fat.empty()) memcpy(fat.data(), d.data(), fat.size() * sizeof(Fat));
        if (fat.size() != slim.size() || recomputeFatPoints) fat.clear();
    }
    if (fat.empty() && !slim.empty()) {
        std::vector<int32_t> triSurfL, cand;
        triSurfL.assign(tris.size() / 9, 0);
        for (size_t s = 0; s < surfaces.size(); ++s)
            for (int t = surfaces[s].triStart; t < surfaces[s].triStart + surfaces[s].triCount; ++t) triSurfL[(size_t)t] = (int32_t)s;
        // the vertical rays of this step (three per spline point) through a bucket grid over the triangles' xz boxes: the triangles of
        // a ray's bucket, in ascending order, are every triangle the scan over all of them could hit, met in the same order -- a
        // 13 k-point spline over 80 k triangles is 3 G triangle tests without it
        struct Buckets {
            float mnx = 0, mnz = 0, cell = 8.0f; int nx = 0, nz = 0;
            std::vector<int32_t> start, ids;
        } bk;
        if (!tris.empty()) {
            float mxx = tris[0], mxz = tris[2]; bk.mnx = tris[0]; bk.mnz = tris[2];
            for (size_t v = 0; v < tris.size(); v += 3) { bk.mnx = std::min(bk.mnx, tris[v]); mxx = std::max(mxx, tris[v]); bk.mnz = std::min(bk.mnz, tris[v + 2]); mxz = std::max(mxz, tris[v + 2]); }
            while ((double)((mxx - bk.mnx) / bk.cell + 1.0f) * (double)((mxz - bk.mnz) / bk.cell + 1.0f) > 4194304.0) bk.cell *= 2.0f;
            bk.nx = (int)floorf((mxx - bk.mnx) / bk.cell) + 1; bk.nz = (int)floorf((mxz - bk.mnz) / bk.cell) + 1;
            const size_t nc = (size_t)bk.nx * (size_t)bk.nz;
            bk.start.assign(nc + 1, 0);
            const float e = 1.0e-3f;
            auto span = [&](size_t t, int& x0, int& x1, int& z0, int& z1) {
                const float* p = tris.data() + 9 * t;
                x0 = std::max((int)floorf((std::min(p[0], std::min(p[3], p[6])) - e - bk.mnx) / bk.cell), 0); x1 = std::min((int)floorf((std::max(p[0], std::max(p[3], p[6])) + e - bk.mnx) / bk.cell), bk.nx - 1);
                z0 = std::max((int)floorf((std::min(p[2], std::min(p[5], p[8])) - e - bk.mnz) / bk.cell), 0); z1 = std::min((int)floorf((std::max(p[2], std::max(p[5], p[8])) + e - bk.mnz) / bk.cell), bk.nz - 1);
            };
            const size_t nt = tris.size() / 9;
            for (size_t t = 0; t < nt; ++t) { int x0, x1, z0, z1; span(t, x0, x1, z0, z1); for (int z = z0; z <= z1; ++z) for (int x = x0; x <= x1; ++x) bk.start[(size_t)z * bk.nx + x + 1]++; }
            for (size_t c = 0; c < nc; ++c) bk.start[c + 1] += bk.start[c];
            bk.ids.assign((size_t)bk.start[nc], 0);
            std::vector<int32_t> fill(bk.start.begin(), bk.start.end() - 1);
            for (size_t t = 0; t < nt; ++t) { int x0, x1, z0, z1; span(t, x0, x1, z0, z1); for (int z = z0; z <= z1; ++z) for (int x = x0; x <= x1; ++x) bk.ids[(size_t)fill[(size_t)z * bk.nx + x]++] = (int32_t)t; }
        }
        std::vector<int32_t> vcand;
        auto rayDown = [&](const V3& o) -> RayHitH {   // rayCastRaw for the direction (0, -1, 0)
            const float dn[3] = {0, -1, 0};
            const int ix = (int)floorf((o.x - bk.mnx) / bk.cell), iz = (int)floorf((o.z - bk.mnz) / bk.cell);
            if (bk.nx == 0 || ix < 0 || iz < 0 || ix >= bk.nx || iz >= bk.nz) return RayHitH();
            const size_t c = (size_t)iz * bk.nx + ix;
            vcand.assign(bk.ids.begin() + bk.start[c], bk.ids.begin() + bk.start[c + 1]);
            return rayCastSubset(vcand, triSurfL, tris.data(), &o.x, dn, traceRayLength);
        };
        const int numTraceSteps = (int)(traceSideMax / traceStep);
        // Track::computeSideLocation (Track.cpp:435-467): fan of rays from the point's ray origin towards origHit + dir*k*step;
        // the side moves outwards while the hit stays on valid track of the same category, within the height / grip steps
        // and outside the bad sectors; a miss is skipped, the first disqualified hit ends the trace.
        auto sideLocation = [&](const RayHitH& orig, const V3& rayStart, const V3& traceDir) {
            const V3 origPos = {orig.pos[0], orig.pos[1], orig.pos[2]};
            V3 result = origPos, prevHit = origPos;
            float prevGrip = surfaces[(size_t)orig.surface].gripMod;
            // horizontal reach of the longest ray: direction (k*step along traceDir, down to the hit), length traceRayLength
            const V3 far = norm(origPos + traceDir * ((float)numTraceSteps * traceStep) - rayStart) * traceRayLength;
            trisNearSegment(tris, rayStart.x, rayStart.z, rayStart.x + far.x * 1.01f, rayStart.z + far.z * 1.01f, cand);
            for (int k = 1; k < numTraceSteps; ++k) {
                const V3 rayEnd = origPos + traceDir * ((float)k * traceStep);
                const V3 rayN = norm(rayEnd - rayStart);
                const RayHitH h = rayCastSubset(cand, triSurfL, tris.data(), &rayStart.x, &rayN.x, traceRayLength);
                if (!h.has) continue;
                const pdb_surface& sf = surfaces[(size_t)h.surface];
                if (sf.isValidTrack && sf.collisionCategory == surfaces[(size_t)orig.surface].collisionCategory &&
                    fabsf(h.pos[1] - prevHit.y) < traceDiffHeightMax && fabsf(sf.gripMod - prevGrip) < traceDiffGripMax &&
                    std::find(traceBadSectors.begin(), traceBadSectors.end(), sf.sectorID) == traceBadSectors.end()) {
                    result = {h.pos[0], h.pos[1], h.pos[2]};
                    prevHit = result;
                    prevGrip = sf.gripMod;
                } else break;
            }
            return result;
        };
        fat.resize(slim.size());
        memset(fat.data(), 0, fat.size() * sizeof(Fat));
        for (size_t i = 0; i < slim.size(); ++i) {
            const V3 sb = {slim[i].best[0], slim[i].best[1], slim[i].best[2]};
            const V3 rs = sb + V3{0, traceRayOffsetY, 0};
            RayHitH h = rayDown(rs);
            if (!h.has) continue;
            Fat& f = fat[i];
            f.best = {h.pos[0], h.pos[1], h.pos[2]};
            if (i + 1 < slim.size()) f.forwardDir = norm(V3{slim[i + 1].best[0], slim[i + 1].best[1], slim[i + 1].best[2]} - sb);
            else if (i > 0) f.forwardDir = norm(sb - V3{slim[i - 1].best[0], slim[i - 1].best[1], slim[i - 1].best[2]});
            const V3 leftDir = norm(cross(f.forwardDir, V3{0, -1, 0}));
            const V3 rightDir = leftDir * -1.0f;
            if (traceSides) {
                f.left = sideLocation(h, rs, leftDir);
                f.right = sideLocation(h, rs, rightDir);
                f.center = (f.left + f.right) * 0.5f;
                continue;
            }
            f.left = f.best + leftDir * slim[i].sides[0];
            f.right = f.best + rightDir * slim[i].sides[1];
            V3 o = f.left + V3{0, traceRayOffsetY, 0};
            h = rayDown(o);
            if (h.has) f.left = {h.pos[0], h.pos[1], h.pos[2]};
            o = f.right + V3{0, traceRayOffsetY, 0};
            h = rayDown(o);
            if (h.has) f.right = {h.pos[0], h.pos[1], h.pos[2]};
            f.center = (f.left + f.right) * 0.5f;
        }
    }

    // ---- initTrackPoints (Track.cpp:200-271) ----
    float trackWidth = 0.1f, trackLength = 0.1f;
    std::vector<float> fatDist(fat.size());
    std::vector<V3> nodes;
    std::vector<float> nodeDist;
    int steps = 0;
    if (!fat.empty()) {
        const size_t n = fat.size();
        for (size_t id = 0; id < n; ++id) {
            const float w = len(fat[id].left - fat[id].right);
            if (trackWidth < w) trackWidth = w;
            fatDist[id] = trackLength;
            if (id + 1 < n) trackLength += len(fat[id].best - fat[id + 1].best);
        }
        steps = (int)(trackLength / 0.1f) / (int)n;
        auto addNode = [&](V3 p) {
            nodes.push_back(p);
            if (nodes.size() == 1) nodeDist.push_back(0);
            else { const size_t k = nodes.size() - 1; nodeDist.push_back(len(nodes[k] - nodes[k - 1]) + nodeDist[k - 1]); }
        };
        const int np = (int)n;
        if (np >= 4) {
            std::vector<V3> pts(n);
            for (size_t i = 0; i < n; ++i) pts[i] = fat[i].best;
            auto wrap = [&](int id) { return id < np ? id : id - np; };
            if (!closedLoop) {
                const float d0 = len(pts[1] - pts[0]);
                const V3 n0 = (pts[1] - pts[0]) / d0;
                for (int i = 0; i < steps; ++i) { const float u = (float)i / (float)steps; addNode(pts[0] + n0 * (u * d0)); }
            }
            for (int pt = 0; pt + 4 < np; ++pt)
                for (int i = 0; i < steps; ++i) { const float u = (float)i / (float)steps; addNode(bsInterp(u, pts[pt], pts[pt + 1], pts[pt + 2], pts[pt + 3])); }
            if (closedLoop) {
                for (int pt = np - 4; pt < np; ++pt)
                    for (int i = 0; i < steps; ++i) {
                        const float u = (float)i / (float)steps;
                        addNode(bsInterp(u, pts[pt], pts[wrap(pt + 1)], pts[wrap(pt + 2)], pts[wrap(pt + 3)]));
                    }
            } else {
                for (int pt = np - 3; pt + 1 < np; ++pt) {
                    const float dx = len(pts[pt + 1] - pts[pt]);
                    const V3 nx = (pts[pt + 1] - pts[pt]) / dx;
                    for (int i = 0; i < steps; ++i) { const float u = (float)i / (float)steps; addNode(pts[pt] + nx * (u * dx)); }
                }
                addNode(pts[np - 1]);
            }
        }
        if (nodeDist.empty()) throw std::runtime_error("pdb: spline needs at least 4 points");
        trackLength = nodeDist.back();
    }

    // ---- pit boxes (Track::loadPits, Track.cpp:151-175): sections AC_PIT_0, AC_PIT_1, ... until the first that is missing; the box's matrix is the
    //      rotation by ROT.x degrees about +y (mat44f::createFromAxisAngle, Core/Math.cpp:88-118, with the C library's sinf / cosf: a load-time
    //      constant like the reference's) with POS in its fourth row ----
    std::vector<float> pits;
    {
        Ini pitIni(dir + "pits.ini");
        if (pitIni.ready) {
            for (int i = 0; ; ++i) {
                const std::string sec = "AC_PIT_" + std::to_string(i);
                if (!pitIni.hasSection(sec)) break;
                float vPos[3], vRot[3];
                pitIni.getFloat3(sec, "POS", vPos);
                pitIni.getFloat3(sec, "ROT", vRot);
                const float ax = 0.0f, ay = 1.0f, az = 0.0f;
                const float angle = vRot[0] * 0.01745329251994329576923690768489f;
                const float sin_a = sinf(angle), cos_a = cosf(angle), om = 1.0f - cos_a;
                float m[16];
                m[0] = ((ax * ax) * om) + cos_a; m[5] = ((ay * ay) * om) + cos_a; m[10] = ((az * az) * om) + cos_a;
                m[1] = (az * sin_a) + (ay * ax) * om; m[6] = (ax * sin_a) + (az * ay) * om; m[8] = (ay * sin_a) + (az * ax) * om;
                m[2] = (az * ax) * om - (ay * sin_a); m[4] = (ay * ax) * om - (az * sin_a); m[9] = (az * ay) * om - (ax * sin_a);
                m[3] = 0; m[7] = 0; m[11] = 0;
                m[12] = vPos[0]; m[13] = vPos[1]; m[14] = vPos[2]; m[15] = 1;
                pits.insert(pits.end(), m, m + 16);
            }
        }
    }

    // ---- pack ----
    pdb_track_header h;
    memset(&h, 0, sizeof(h));
    h.magic = 0x4B544450; h.version = 6;
    h.numPits = (int32_t)(pits.size() / 16);
    h.numSurfaces = (int32_t)surfaces.size(); h.numTris = (int32_t)(tris.size() / 9);
    h.numFat = (int32_t)fat.size(); h.numNodes = (int32_t)nodes.size();
    h.interpolateStep = steps; h.closedLoop = closedLoop ? 1 : 0;
    h.computedTrackLength = trackLength; h.computedTrackWidth = trackWidth; h.dynamicGripLevel = grip; h.hashCellSize = cellSize;
    auto align = [](uint64_t x) { return (x + 15) & ~uint64_t(15); };
    uint64_t off = align(sizeof(h));
    h.offSurfaces = off; off = align(off + surfaces.size() * sizeof(pdb_surface));
    h.offTris = off; off = align(off + tris.size() * 4);
    h.offFat = off; off = align(off + fat.size() * 60);
    h.offFatDist = off; off = align(off + fat.size() * 4);
    h.offNodes = off; off = align(off + nodes.size() * 12);
    h.offNodeDist = off; off = align(off + nodes.size() * 4);
    // ---- xz grid over the triangles (see pdb_track_header) ----
    std::vector<int32_t> gridStart, gridTris, triSurf((size_t)h.numTris, 0);
    for (size_t s = 0; s < surfaces.size(); ++s)
        for (int t = surfaces[s].triStart; t < surfaces[s].triStart + surfaces[s].triCount; ++t) triSurf[(size_t)t] = (int32_t)s | (surfaces[s].collisionCategory << 24);
    if (h.numTris > 0) {
        float mnx = tris[0], mxx = tris[0], mnz = tris[2], mxz = tris[2];
        for (size_t v = 0; v < tris.size(); v += 3) { mnx = std::min(mnx, tris[v]); mxx = std::max(mxx, tris[v]); mnz = std::min(mnz, tris[v + 2]); mxz = std::max(mxz, tris[v + 2]); }
        // cell edge: start from "two triangles per cell of the bounding box", then -- a road is a thin ribbon in a mostly empty box --
        // halve it while the cells that hold anything hold more than three list entries on average (not below 1 m, not beyond 2^19 cells).
        // (The grid only selects candidates: any cell size gives the same hits.)
        float cell = sqrtf(((mxx - mnx) * (mxz - mnz)) / (float)h.numTris * 2.0f);
        if (!(cell > 0.5f)) cell = 0.5f;
        auto cellsAt = [&](float c) { return (double)((mxx - mnx) / c + 1.0f) * (double)((mxz - mnz) / c + 1.0f); };
        while (cellsAt(cell) > 1048576.0) cell *= 2.0f;
        for (;;) {
            const float half = cell * 0.5f;
            if (half < 1.0f || cellsAt(half) > 524288.0) break;
            const int nx = (int)floorf((mxx - mnx) / cell) + 1, nz = (int)floorf((mxz - mnz) / cell) + 1;
            std::vector<int32_t> cnt((size_t)nx * (size_t)nz, 0);
            size_t entries = 0, nonEmpty = 0;
            for (int t = 0; t < h.numTris; ++t) {
                const float* p = tris.data() + 9 * (size_t)t;
                const int x0 = (int)floorf((std::min(p[0], std::min(p[3], p[6])) - mnx) / cell), x1 = (int)floorf((std::max(p[0], std::max(p[3], p[6])) - mnx) / cell);
                const int z0 = (int)floorf((std::min(p[2], std::min(p[5], p[8])) - mnz) / cell), z1 = (int)floorf((std::max(p[2], std::max(p[5], p[8])) - mnz) / cell);
                for (int z = z0; z <= z1; ++z) for (int x = x0; x <= x1; ++x) { if (cnt[(size_t)z * nx + x]++ == 0) ++nonEmpty; ++entries; }
            }
            if (nonEmpty == 0 || (double)entries / (double)nonEmpty <= 3.0) break;
            cell = half;
        }
        h.gridMinX = mnx; h.gridMinZ = mnz; h.gridCell = cell;
        auto cellOf = [&](float x, float mn) { return (int)floorf((x - mn) / cell); };   // the kernel uses the same expression
        h.gridNx = cellOf(mxx, mnx) + 1; h.gridNz = cellOf(mxz, mnz) + 1;
        const size_t nc = (size_t)h.gridNx * (size_t)h.gridNz;
        std::vector<int32_t> count(nc + 1, 0);
        auto span = [&](int t, int& x0, int& x1, int& z0, int& z1) {
            const float* p = tris.data() + 9 * (size_t)t;
            x0 = cellOf(std::min(p[0], std::min(p[3], p[6])), mnx); x1 = cellOf(std::max(p[0], std::max(p[3], p[6])), mnx);
            z0 = cellOf(std::min(p[2], std::min(p[5], p[8])), mnz); z1 = cellOf(std::max(p[2], std::max(p[5], p[8])), mnz);
        };
        for (int t = 0; t < h.numTris; ++t) { int x0, x1, z0, z1; span(t, x0, x1, z0, z1); for (int z = z0; z <= z1; ++z) for (int x = x0; x <= x1; ++x) count[(size_t)z * h.gridNx + x + 1]++; }
        for (size_t c = 0; c < nc; ++c) count[c + 1] += count[c];
        gridStart = count;
        gridTris.assign((size_t)gridStart[nc], 0);
        std::vector<int32_t> fill(gridStart.begin(), gridStart.end() - 1);
        for (int t = 0; t < h.numTris; ++t) { int x0, x1, z0, z1; span(t, x0, x1, z0, z1); for (int z = z0; z <= z1; ++z) for (int x = x0; x <= x1; ++x) gridTris[(size_t)fill[(size_t)z * h.gridNx + x]++] = t; }   // ascending t per cell
    }
    h.offGridStart = off; off = align(off + gridStart.size() * 4);
    h.offGridTris = off; off = align(off + gridTris.size() * 4);
    h.offTriSurf = off; off = align(off + triSurf.size() * 4);
    // ---- xz grid over the fat points (see pdb_track_header) ----
    std::vector<int32_t> fgStart, fgIds;
    if (!fat.empty()) {
        float mnx = fat[0].best.x, mxx = mnx, mnz = fat[0].best.z, mxz = mnz;
        for (const Fat& f : fat) { mnx = std::min(mnx, f.best.x); mxx = std::max(mxx, f.best.x); mnz = std::min(mnz, f.best.z); mxz = std::max(mxz, f.best.z); }
        float cell = cellSize > 1.0f ? cellSize : 50.0f;
        while ((double)((mxx - mnx) / cell + 1.0f) * (double)((mxz - mnz) / cell + 1.0f) > 1048576.0) cell *= 2.0f;
        h.fatGridMinX = mnx; h.fatGridMinZ = mnz; h.fatGridCell = cell;
        auto cellOf = [&](float x, float mn) { return (int)floorf((x - mn) / cell); };   // the kernel uses the same expression
        h.fatGridNx = cellOf(mxx, mnx) + 1; h.fatGridNz = cellOf(mxz, mnz) + 1;
        const size_t nc = (size_t)h.fatGridNx * (size_t)h.fatGridNz;
        fgStart.assign(nc + 1, 0);
        for (const Fat& f : fat) fgStart[(size_t)cellOf(f.best.z, mnz) * h.fatGridNx + cellOf(f.best.x, mnx) + 1]++;
        for (size_t c = 0; c < nc; ++c) fgStart[c + 1] += fgStart[c];
        fgIds.assign(fat.size(), 0);
        std::vector<int32_t> fill(fgStart.begin(), fgStart.end() - 1);
        for (size_t i = 0; i < fat.size(); ++i) fgIds[(size_t)fill[(size_t)cellOf(fat[i].best.z, mnz) * h.fatGridNx + cellOf(fat[i].best.x, mnx)]++] = (int32_t)i;   // ascending id per cell
    }
    h.offFatGridStart = off; off = align(off + fgStart.size() * 4);
    h.offFatGridIds = off; off = align(off + fgIds.size() * 4);
    std::vector<float> fgRec(fgIds.size() * 4, 0.0f), fatSeg(fat.size() * 8, 0.0f);
    for (size_t k = 0; k < fgIds.size(); ++k) {
        const Fat& f = fat[(size_t)fgIds[k]];
        fgRec[4 * k] = f.best.x; fgRec[4 * k + 1] = f.best.y; fgRec[4 * k + 2] = f.best.z;
        memcpy(&fgRec[4 * k + 3], &fgIds[k], 4);
    }
    for (size_t i = 0; i < fat.size(); ++i) {
        const Fat& a = fat[i]; const Fat& b = fat[(i + 1 < fat.size()) ? i + 1 : 0];
        float* q = &fatSeg[8 * i];
        q[0] = a.left.x; q[1] = a.left.z; q[2] = b.left.x; q[3] = b.left.z; q[4] = a.right.x; q[5] = a.right.z; q[6] = b.right.x; q[7] = b.right.z;
    }
    h.offFatGridRec = off; off = align(off + fgRec.size() * 4);
    h.offFatSeg = off; off = align(off + fatSeg.size() * 4);
    // ---- the wheel rays' grid (see pdb_track_header): exact (conservatively inflated) triangle / cell overlap in xz ----
    std::vector<int32_t> rayStart;
    std::vector<pdb_ray_rec> rayRecs;
    if (h.numTris > 0) {
        double mnx = tris[0], mxx = tris[0], mnz = tris[2], mxz = tris[2], mxa = 0.0;
        for (size_t v = 0; v < tris.size(); v += 3) {
            mnx = std::min(mnx, (double)tris[v]); mxx = std::max(mxx, (double)tris[v]); mnz = std::min(mnz, (double)tris[v + 2]); mxz = std::max(mxz, (double)tris[v + 2]);
            mxa = std::max(mxa, std::max(std::fabs((double)tris[v]), std::fabs((double)tris[v + 2])));
        }
        // cell edge: a quarter of the collision grid's, not below half a metre, at most 2^22 cells
        float cell = h.gridCell * 0.25f;
        if (!(cell > 0.5f)) cell = 0.5f;
        auto cellsAt = [&](float c) { return ((mxx - mnx) / c + 1.0) * ((mxz - mnz) / c + 1.0); };
        while (cellsAt(cell) > 4194304.0) cell *= 2.0f;
        h.rayMinX = (float)mnx; h.rayMinZ = (float)mnz; h.rayCell = cell;
        const float fmnx = h.rayMinX, fmnz = h.rayMinZ;
        auto cellOf = [&](float x, float mn) { return (int)floorf((x - mn) / cell); };   // the kernel uses the same expression
        h.rayNx = cellOf((float)mxx, fmnx) + 1; h.rayNz = cellOf((float)mxz, fmnz) + 1;
        // a ray's cell comes out of float arithmetic and so do the inside tests of the hit: every rectangle is inflated by far more
        // than either can be off (a thousandth of a cell plus 2^-17 of the largest coordinate)
        const double eps = 1.0e-3 * (double)cell + 7.62939453125e-6 * mxa;
        auto overlaps = [&](const float* p, int x, int z) {
            const double rx0 = (double)fmnx + (double)x * cell - eps, rx1 = (double)fmnx + (double)(x + 1) * cell + eps;
            const double rz0 = (double)fmnz + (double)z * cell - eps, rz1 = (double)fmnz + (double)(z + 1) * cell + eps;
            const double tx[3] = {p[0], p[3], p[6]}, tz[3] = {p[2], p[5], p[8]};
            for (int e = 0; e < 3; ++e) {   // separating axis = an edge's normal (the box axes are covered by the cell span)
                const int f = (e + 1) % 3, g = (e + 2) % 3;
                const double nx = -(tz[f] - tz[e]), nz = tx[f] - tx[e];
                if (nx == 0.0 && nz == 0.0) continue;
                const double side = nx * (tx[g] - tx[e]) + nz * (tz[g] - tz[e]);   // the triangle lies on this side of the edge (or on it)
                const double c0 = nx * (rx0 - tx[e]) + nz * (rz0 - tz[e]), c1 = nx * (rx1 - tx[e]) + nz * (rz0 - tz[e]);
                const double c2 = nx * (rx0 - tx[e]) + nz * (rz1 - tz[e]), c3 = nx * (rx1 - tx[e]) + nz * (rz1 - tz[e]);
                if (side >= 0.0 ? (c0 < 0.0 && c1 < 0.0 && c2 < 0.0 && c3 < 0.0) : (c0 > 0.0 && c1 > 0.0 && c2 > 0.0 && c3 > 0.0)) return false;
            }
            return true;
        };
        auto span = [&](int t, int& x0, int& x1, int& z0, int& z1) {
            const float* p = tris.data() + 9 * (size_t)t;
            const float e = (float)eps;
            x0 = std::max(cellOf(std::min(p[0], std::min(p[3], p[6])) - e, fmnx), 0); x1 = std::min(cellOf(std::max(p[0], std::max(p[3], p[6])) + e, fmnx), h.rayNx - 1);
            z0 = std::max(cellOf(std::min(p[2], std::min(p[5], p[8])) - e, fmnz), 0); z1 = std::min(cellOf(std::max(p[2], std::max(p[5], p[8])) + e, fmnz), h.rayNz - 1);
        };
        const size_t nc = (size_t)h.rayNx * (size_t)h.rayNz;
        rayStart.assign(nc + 1, 0);
        std::vector<std::pair<int32_t, int32_t>> ent;   // (cell, triangle), triangles ascending
        for (int t = 0; t < h.numTris; ++t) {
            int x0, x1, z0, z1; span(t, x0, x1, z0, z1);
            const float* p = tris.data() + 9 * (size_t)t;
            for (int z = z0; z <= z1; ++z) for (int x = x0; x <= x1; ++x)
                if (overlaps(p, x, z)) { ent.emplace_back((int32_t)((size_t)z * h.rayNx + x), (int32_t)t); rayStart[(size_t)z * h.rayNx + x + 1]++; }
        }
        for (size_t c = 0; c < nc; ++c) rayStart[c + 1] += rayStart[c];
        rayRecs.resize(ent.size());
        std::vector<int32_t> fill(rayStart.begin(), rayStart.end() - 1);
        for (const auto& e : ent) {
            pdb_ray_rec& r = rayRecs[(size_t)fill[(size_t)e.first]++];
            memcpy(r.v, tris.data() + 9 * (size_t)e.second, 36);
            r.tri = e.second; r.surface = triSurf[(size_t)e.second] & 0xFFFFFF; r._pad = 0;
        }
    }
    h.offRayStart = off; off = align(off + rayStart.size() * 4);
    h.offRayRecs = off; off = align(off + rayRecs.size() * sizeof(pdb_ray_rec));
    h.offPits = off; off = align(off + pits.size() * 4);
    h.totalBytes = off;
    std::vector<uint8_t> blob(off, 0);
    memcpy(blob.data(), &h, sizeof(h));
    if (!surfaces.empty()) memcpy(blob.data() + h.offSurfaces, surfaces.data(), surfaces.size() * sizeof(pdb_surface));
    if (!tris.empty()) memcpy(blob.data() + h.offTris, tris.data(), tris.size() * 4);
    if (!fat.empty()) memcpy(blob.data() + h.offFat, fat.data(), fat.size() * 60);
    if (!fat.empty()) memcpy(blob.data() + h.offFatDist, fatDist.data(), fat.size() * 4);
    if (!nodes.empty()) memcpy(blob.data() + h.offNodes, nodes.data(), nodes.size() * 12);
    if (!nodes.empty()) memcpy(blob.data() + h.offNodeDist, nodeDist.data(), nodes.size() * 4);
    if (!gridStart.empty()) memcpy(blob.data() + h.offGridStart, gridStart.data(), gridStart.size() * 4);
    if (!gridTris.empty()) memcpy(blob.data() + h.offGridTris, gridTris.data(), gridTris.size() * 4);
    if (!triSurf.empty()) memcpy(blob.data() + h.offTriSurf, triSurf.data(), triSurf.size() * 4);
    if (!fgStart.empty()) memcpy(blob.data() + h.offFatGridStart, fgStart.data(), fgStart.size() * 4);
    if (!fgIds.empty()) memcpy(blob.data() + h.offFatGridIds, fgIds.data(), fgIds.size() * 4);
    if (!fgRec.empty()) memcpy(blob.data() + h.offFatGridRec, fgRec.data(), fgRec.size() * 4);
    if (!fatSeg.empty()) memcpy(blob.data() + h.offFatSeg, fatSeg.data(), fatSeg.size() * 4);
    if (!rayStart.empty()) memcpy(blob.data() + h.offRayStart, rayStart.data(), rayStart.size() * 4);
    if (!rayRecs.empty()) memcpy(blob.data() + h.offRayRecs, rayRecs.data(), rayRecs.size() * sizeof(pdb_ray_rec));
    if (!pits.empty()) memcpy(blob.data() + h.offPits, pits.data(), pits.size() * 4);
    return blob;
}

}  // namespace pdb
