// ORACLE / TEST INFRASTRUCTURE.  Contact generation between a car's colliders and the static track triangles.
//
// PARITY UNPINNED, OWN DEFINITION: in the reference this is ODE's dCollide (box-trimesh, trimesh-trimesh), reached from
// PhysicsEngineODE::collisionNearCallback (Physics/ODE/PhysicsEngineODE.cpp:246-281); ODE is not in the reference tree.
// What is defined here (and mirrored by the HIP kernel, bit for bit):
//   * box vs triangle: does the box reach the triangle's plane (world-space evaluation), then the 13-axis separating-axis
//     test in the box frame; one contact per intersecting triangle, normal = the triangle's turned towards the box centre;
//   * hull vs triangle: every edge of one triangle that pierces the other is a contact at the piercing point, normal = the
//     static triangle's turned towards the body origin.  Two rejections come first, both part of the definition (the kernel
//     evaluates the very same expressions): a wall triangle that does not meet the colliders' bounding box (separating-axis
//     test in the body frame) meets nothing; a pair (hull triangle, wall triangle) where one triangle's vertices lie strictly
//     on one side of the other's plane has no piercing edge.  They make the work proportional to what actually touches: the
//     reference's tracks put hundreds of centimetre-sized wall triangles inside a car's bounding box (driftplayground).
//   * contact POINTS for the response (the joints PhysicsEngineODE::onCollision creates, :283-331), each with a depth:
//       box: every box corner behind the triangle's plane whose projection falls inside the triangle (depth = distance behind
//            the plane) and, per triangle edge, the midpoint of the part of the edge inside the box (depth = distance from there
//            to where the ray against the normal leaves the box);
//       hull: the piercing points, depth = how far the piercing hull edge's deeper end (hull edges) or the pierced hull
//            triangle's deepest vertex (wall edges) lies behind the wall triangle's plane.
//     A car keeps its PDB_MAX_CONTACTS deepest contact points (ties: lowest id = triangle * 2048 + item), which makes the kept
//     set independent of the order in which triangles are visited (ODE keeps the first 32 per geom pair in ITS order).
// Used by the CPU restatement (cpu_ref: feeds its restatement of Car::onCollisionCallback) and by the fixture harness's
// engine (refharness: feeds the REFERENCE's own Simulator/Car::onCollisionCallback), so that what the reference does with
// a contact is pinned even though where contacts come from is not.
#pragma once
#include <cmath>

namespace pdcol {

struct V { float x, y, z; };
static inline V mk(float x, float y, float z) { V v; v.x = x; v.y = y; v.z = z; return v; }
static inline V ld(const float* p) { return mk(p[0], p[1], p[2]); }
static inline V operator+(const V& a, const V& b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V operator-(const V& a, const V& b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V operator*(const V& a, float f) { return mk(a.x * f, a.y * f, a.z * f); }
static inline float dot(const V& a, const V& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V cross(const V& a, const V& v) { return mk(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x); }
static inline float len(const V& a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline V norm(const V& a) { const float l = len(a); if (l != 0.0f) { const float s = 1.0f / l; return mk(a.x * s, a.y * s, a.z * s); } return a; }
static inline float tmin(float a, float b) { return a < b ? a : b; }
static inline float tmax(float a, float b) { return a > b ? a : b; }

// rigid pose: world = pos + R * local, R row-major 3x3 (dBodyGetRelPointPos / dBodyGetPosRelPoint)
struct Pose { float pos[3]; float R[9]; };
static inline V toWorld(const Pose& P, const V& p) {
    const float* R = P.R;
    return mk((R[0] * p.x + R[1] * p.y + R[2] * p.z) + P.pos[0], (R[3] * p.x + R[4] * p.y + R[5] * p.z) + P.pos[1], (R[6] * p.x + R[7] * p.y + R[8] * p.z) + P.pos[2]);
}
static inline V toLocal(const Pose& P, const V& p) {
    const float* R = P.R;
    const V d = mk(p.x - P.pos[0], p.y - P.pos[1], p.z - P.pos[2]);
    return mk(R[0] * d.x + R[3] * d.y + R[6] * d.z, R[1] * d.x + R[4] * d.y + R[7] * d.z, R[2] * d.x + R[5] * d.y + R[8] * d.z);
}

// segment a->b against triangle p0 p1 p2, either side; the piercing point
static inline bool segTri(const V& a, const V& b, const V& p0, const V& p1, const V& p2, V& hit) {
    const V e1 = p1 - p0, e2 = p2 - p0, d = b - a;
    const V pv = cross(d, e2);
    float det = dot(e1, pv);
    const V tv = a - p0;
    float u = dot(tv, pv);
    const V qv = cross(tv, e1);
    float v = dot(d, qv);
    float t = dot(e2, qv);
    if (det < 0.0f) { det = -det; u = -u; v = -v; t = -t; }
    if (!(det > 0.0f)) return false;
    if (u < 0.0f || v < 0.0f || u + v > det || t < 0.0f || t > det) return false;
    const float sc = t / det;
    hit = mk(a.x + d.x * sc, a.y + d.y * sc, a.z + d.z * sc);
    return true;
}
// box (half extents h, centred at the origin, axis-aligned in its own frame) against triangle q0 q1 q2 given in that frame;
// normalY = y of the unit triangle normal turned towards the box centre
static inline bool boxTri(const V& h, const V& q0, const V& q1, const V& q2, float& normalY) {
    const V f0 = q1 - q0, f1 = q2 - q1, f2 = q0 - q2;
    const V n = cross(f0, q2 - q0);
    const float d = dot(n, q0);
    if (fabsf(d) > h.x * fabsf(n.x) + h.y * fabsf(n.y) + h.z * fabsf(n.z)) return false;
    if (tmin(q0.x, tmin(q1.x, q2.x)) > h.x || tmax(q0.x, tmax(q1.x, q2.x)) < -h.x) return false;
    if (tmin(q0.y, tmin(q1.y, q2.y)) > h.y || tmax(q0.y, tmax(q1.y, q2.y)) < -h.y) return false;
    if (tmin(q0.z, tmin(q1.z, q2.z)) > h.z || tmax(q0.z, tmax(q1.z, q2.z)) < -h.z) return false;
    const V f[3] = {f0, f1, f2};
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
            const V a = (k == 0) ? mk(0.0f, -f[j].z, f[j].y) : (k == 1) ? mk(f[j].z, 0.0f, -f[j].x) : mk(-f[j].y, f[j].x, 0.0f);   // e_k x f_j
            const float p0 = dot(a, q0), p1 = dot(a, q1), p2 = dot(a, q2);
            const float r = h.x * fabsf(a.x) + h.y * fabsf(a.y) + h.z * fabsf(a.z);
            if (tmin(p0, tmin(p1, p2)) > r || tmax(p0, tmax(p1, p2)) < -r) return false;
        }
    const float l = len(n);
    float ny = (l != 0.0f) ? n.y / l : 0.0f;
    if (d > 0.0f) ny = -ny;
    normalY = ny;
    return true;
}

// world AABB of a body-frame box [lo, hi] under pose P
static inline void worldAabb(const Pose& P, const float* lo, const float* hi, V& aLo, V& aHi) {
    const V l = ld(lo), h = ld(hi);
    const V cb = (l + h) * 0.5f, hb = (h - l) * 0.5f;
    const V cw = toWorld(P, cb);
    const float* R = P.R;
    const V ext = mk(fabsf(R[0]) * hb.x + fabsf(R[1]) * hb.y + fabsf(R[2]) * hb.z, fabsf(R[3]) * hb.x + fabsf(R[4]) * hb.y + fabsf(R[5]) * hb.z,
                     fabsf(R[6]) * hb.x + fabsf(R[7]) * hb.y + fabsf(R[8]) * hb.z);
    aLo = cw - ext; aHi = cw + ext;
}
static inline bool triMeetsAabb(const V& p0, const V& p1, const V& p2, const V& aLo, const V& aHi) {
    if (tmin(p0.x, tmin(p1.x, p2.x)) > aHi.x || tmax(p0.x, tmax(p1.x, p2.x)) < aLo.x) return false;
    if (tmin(p0.y, tmin(p1.y, p2.y)) > aHi.y || tmax(p0.y, tmax(p1.y, p2.y)) < aLo.y) return false;
    if (tmin(p0.z, tmin(p1.z, p2.z)) > aHi.z || tmax(p0.z, tmax(p1.z, p2.z)) < aLo.z) return false;
    return true;
}

// belly box against one triangle: true + body-local normal.y of the contact (PhysicsEngineODE::onCollision keeps >= 0.9)
static inline bool boxContact(const Pose& P, const float* centre, const float* half, const V& p0, const V& p1, const V& p2, float& localNormalY) {
    const V bc = ld(centre), bh = ld(half);
    const V nW = cross(p1 - p0, p2 - p0);
    const float dW = dot(nW, toWorld(P, bc) - p0);
    const float* R = P.R;
    const float rW = bh.x * fabsf(dot(nW, mk(R[0], R[3], R[6]))) + bh.y * fabsf(dot(nW, mk(R[1], R[4], R[7]))) + bh.z * fabsf(dot(nW, mk(R[2], R[5], R[8])));
    if (fabsf(dW) > rW) return false;
    return boxTri(bh, toLocal(P, p0) - bc, toLocal(P, p1) - bc, toLocal(P, p2) - bc, localNormalY);
}

// hull (body-frame vertices, uint8 index triples) against one triangle: emit(normal, pos) per contact
// one contact point = what a dContactGeom carries into dJointCreateContact (layout = pdb_contact, include/pdb_types.h)
struct Contact { float pos[3]; float depth; float normal[3]; int kind; };   // kind 0: hull vs WALL (mode 28692), 1: box vs TRACK (mode 28700)
enum { MAX_CONTACTS = 32, ITEM_BOX_CORNER = 1152, ITEM_BOX_EDGE = 1160, ID_STRIDE = 2048 };
static inline bool contactBefore(float da, unsigned ia, float db, unsigned ib) { return da > db || (da == db && ia < ib); }
struct ContactSet {
    int n = 0;
    Contact c[MAX_CONTACTS];
    unsigned id[MAX_CONTACTS];
    void clear() { n = 0; }
    void insert(const V& pos, const V& nrm, float depth, int kind, unsigned kid) {
        if (!(depth >= 0.0f) || !(depth < 1.0e30f)) return;
        int p = 0;
        while (p < n && contactBefore(c[p].depth, id[p], depth, kid)) ++p;
        if (p >= MAX_CONTACTS) return;
        const int last = n < MAX_CONTACTS ? n : MAX_CONTACTS - 1;
        for (int i = last; i > p; --i) { c[i] = c[i - 1]; id[i] = id[i - 1]; }
        Contact k; k.pos[0] = pos.x; k.pos[1] = pos.y; k.pos[2] = pos.z; k.depth = depth; k.normal[0] = nrm.x; k.normal[1] = nrm.y; k.normal[2] = nrm.z; k.kind = kind;
        c[p] = k; id[p] = kid;
        if (n < MAX_CONTACTS) ++n;
    }
};

// belly box against one triangle, with contact points: returns like boxContact (intersects; body-local normal.y); when the
// normal passes the reference's filter (>= 0.9, PhysicsEngineODE.cpp:303-312) emit(posWorld, normalWorld, depth, item)
template <typename Emit>
static inline bool boxContacts(const Pose& P, const float* centre, const float* half, const V& p0, const V& p1, const V& p2, float& localNormalY, Emit emit) {
    if (!boxContact(P, centre, half, p0, p1, p2, localNormalY)) return false;
    if (!(localNormalY >= 0.9f)) return true;
    const V bc = ld(centre), bh = ld(half);
    const V q0 = toLocal(P, p0) - bc, q1 = toLocal(P, p1) - bc, q2 = toLocal(P, p2) - bc;
    const V n0 = cross(q1 - q0, q2 - q0);
    const float l = len(n0);
    const float inv = 1.0f / l;
    V nh = mk(n0.x * inv, n0.y * inv, n0.z * inv);
    if (dot(n0, q0) > 0.0f) nh = nh * -1.0f;
    const float* R = P.R;
    const V nW = mk(R[0] * nh.x + R[1] * nh.y + R[2] * nh.z, R[3] * nh.x + R[4] * nh.y + R[5] * nh.z, R[6] * nh.x + R[7] * nh.y + R[8] * nh.z);
    for (int k = 0; k < 8; ++k) {
        const V v = mk((k & 1) ? bh.x : -bh.x, (k & 2) ? bh.y : -bh.y, (k & 4) ? bh.z : -bh.z);
        const float s = dot(nh, v - q0);
        if (!(s < 0.0f)) continue;
        const V pp = v - nh * s;
        if (!(dot(cross(q1 - q0, pp - q0), n0) >= 0.0f && dot(cross(q2 - q1, pp - q1), n0) >= 0.0f && dot(cross(q0 - q2, pp - q2), n0) >= 0.0f)) continue;
        emit(toWorld(P, v + bc), nW, -s, (int)ITEM_BOX_CORNER + k);
    }
    const V q[4] = {q0, q1, q2, q0};
    const float hh[3] = {bh.x, bh.y, bh.z};
    for (int j = 0; j < 3; ++j) {
        const V a = q[j], dd = q[j + 1] - q[j];
        const float av[3] = {a.x, a.y, a.z}, dv[3] = {dd.x, dd.y, dd.z};
        float t0 = 0.0f, t1 = 1.0f;
        bool ok = true;
        for (int i = 0; i < 3; ++i) {
            if (dv[i] == 0.0f) { if (av[i] < -hh[i] || av[i] > hh[i]) ok = false; }
            else {
                float ta = (-hh[i] - av[i]) / dv[i], tb = (hh[i] - av[i]) / dv[i];
                if (ta > tb) { const float t = ta; ta = tb; tb = t; }
                if (ta > t0) t0 = ta;
                if (tb < t1) t1 = tb;
            }
        }
        if (!ok || t0 > t1) continue;
        const float tm = (t0 + t1) * 0.5f;
        const V pm = mk(a.x + dd.x * tm, a.y + dd.y * tm, a.z + dd.z * tm);
        const float pv[3] = {pm.x, pm.y, pm.z}, nv[3] = {nh.x, nh.y, nh.z};
        float depth = 3.0e38f;
        for (int i = 0; i < 3; ++i) {
            const float dir = -nv[i];
            float te;
            if (dir > 0.0f) te = (hh[i] - pv[i]) / dir; else if (dir < 0.0f) te = (-hh[i] - pv[i]) / dir; else continue;
            if (te < depth) depth = te;
        }
        emit(toWorld(P, pm + bc), nW, depth, (int)ITEM_BOX_EDGE + j);
    }
    return true;
}

// does the triangle meet the colliders' bounding box [lo, hi] (body frame)?  The hull lies inside it.
static inline bool triMeetsBounds(const Pose& P, const float* lo, const float* hi, const V& p0, const V& p1, const V& p2) {
    const V l = ld(lo), h = ld(hi);
    const V cb = (l + h) * 0.5f, hb = (h - l) * 0.5f;
    float ny;
    return boxTri(hb, toLocal(P, p0) - cb, toLocal(P, p1) - cb, toLocal(P, p2) - cb, ny);
}
static inline bool strictlyOneSide(float a, float b, float c) { return (a > 0.0f && b > 0.0f && c > 0.0f) || (a < 0.0f && b < 0.0f && c < 0.0f); }

template <typename Emit>
static inline void hullContacts(const Pose& P, const float (*verts)[3], const unsigned char (*tris)[3], int numTris, const V& p0, const V& p1, const V& p2, Emit emit) {
    const V nT = cross(p1 - p0, p2 - p0);
    const V nw = norm(nT);
    for (int ct = 0; ct < numTris; ++ct) {
        const V c0 = toWorld(P, ld(verts[tris[ct][0]])), c1 = toWorld(P, ld(verts[tris[ct][1]])), c2 = toWorld(P, ld(verts[tris[ct][2]]));
        if (strictlyOneSide(dot(nT, c0 - p0), dot(nT, c1 - p0), dot(nT, c2 - p0))) continue;   // the hull triangle does not reach the wall triangle's plane
        const V nH = cross(c1 - c0, c2 - c0);
        if (strictlyOneSide(dot(nH, p0 - c0), dot(nH, p1 - c0), dot(nH, p2 - c0))) continue;   // nor the other way round
        for (int e = 0; e < 6; ++e) {
            V hit;
            bool got;
            if (e == 0) got = segTri(c0, c1, p0, p1, p2, hit); else if (e == 1) got = segTri(c1, c2, p0, p1, p2, hit); else if (e == 2) got = segTri(c2, c0, p0, p1, p2, hit);
            else if (e == 3) got = segTri(p0, p1, c0, c1, c2, hit); else if (e == 4) got = segTri(p1, p2, c0, c1, c2, hit); else got = segTri(p2, p0, c0, c1, c2, hit);
            if (!got) continue;
            V n = nw;
            if (dot(n, ld(P.pos) - hit) < 0.0f) n = n * -1.0f;
            // depth behind the wall triangle's plane (through p0, normal n): the piercing hull edge's ends (e < 3), or the
            // pierced hull triangle's vertices (wall edges, e >= 3)
            float m;
            if (e < 3) { const V a = (e == 0) ? c0 : (e == 1) ? c1 : c2, b = (e == 0) ? c1 : (e == 1) ? c2 : c0; m = tmin(dot(n, a - p0), dot(n, b - p0)); }
            else m = tmin(dot(n, c0 - p0), tmin(dot(n, c1 - p0), dot(n, c2 - p0)));
            emit(n, hit, (m < 0.0f) ? -m : 0.0f, ct * 6 + e);
        }
    }
}

}  // namespace pdcol
