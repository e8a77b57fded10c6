// ORACLE / TEST INFRASTRUCTURE.  Ray vs static triangle meshes, restating what the reference gets
// from ODE/OPCODE at PhysicsEngineODE.cpp:154-214 (rayCastImpl + rayNearCallback) and
// RayCasterODE.cpp:12-28 (back-face culling on, one contact per mesh).  PARITY UNPINNED (ODE absent):
//  * per mesh the NEAREST culled hit is kept (OPCODE "first contact" returns traversal-order first
//    hit, which cannot be reproduced without the library; identical for non-overlapping ground);
//  * hit normal = normalise((v1-v0) x (v2-v0)): ODE's trimesh-ray collider writes (v2-v0)x(v1-v0)
//    and dCollide's reversed (ray, trimesh) dispatch negates it.
#pragma once
#include <vector>
#include <cmath>
#include <cstdint>

namespace pdrb {

struct StaticMesh {
    std::vector<float> verts;        // xyz
    std::vector<uint16_t> indices;   // 3 per triangle
    void* user = nullptr;            // Surface*
    unsigned long category = 0, mask = 0;
};

struct RayHit {
    bool has = false;
    float depth = -1.0f;
    float pos[3] = {0, 0, 0};
    float normal[3] = {0, 0, 0};
    int mesh = -1;
};

// Moeller-Trumbore with OPCODE's culling form (det = edge1 . (dir x edge2) must exceed 1e-6).
inline bool rayTri(const float* o, const float* d, float maxDist, const float* v0, const float* v1,
                   const float* v2, float& tOut) {
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

inline RayHit rayCastMeshes(const std::vector<StaticMesh>& meshes, const float* o, const float* d, float maxDist) {
    RayHit best;
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
        const StaticMesh& m = meshes[mi];
        float bt = -1.0f;
        size_t btri = 0;
        const size_t nt = m.indices.size() / 3;
        for (size_t t = 0; t < nt; ++t) {
            const float* v0 = &m.verts[3 * m.indices[3 * t + 0]];
            const float* v1 = &m.verts[3 * m.indices[3 * t + 1]];
            const float* v2 = &m.verts[3 * m.indices[3 * t + 2]];
            float tt;
            if (rayTri(o, d, maxDist, v0, v1, v2, tt)) {
                if (bt < 0.0f || tt < bt) { bt = tt; btri = t; }
            }
        }
        if (bt >= 0.0f) {
            // PhysicsEngineODE.cpp:205-208: keep if no result yet or strictly nearer
            if (best.depth < 0.0f || best.depth > bt) {
                const float* v0 = &m.verts[3 * m.indices[3 * btri + 0]];
                const float* v1 = &m.verts[3 * m.indices[3 * btri + 1]];
                const float* v2 = &m.verts[3 * m.indices[3 * btri + 2]];
                const float vu[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
                const float vv[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
                float n[3] = {vu[1] * vv[2] - vu[2] * vv[1], vu[2] * vv[0] - vu[0] * vv[2], vu[0] * vv[1] - vu[1] * vv[0]};
                const float l = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                if (l > 0.0f) {
                    const float s = 1.0f / sqrtf(l);
                    n[0] *= s; n[1] *= s; n[2] *= s;
                    best.has = true;
                    best.depth = bt;
                    best.pos[0] = o[0] + d[0] * bt;
                    best.pos[1] = o[1] + d[1] * bt;
                    best.pos[2] = o[2] + d[2] * bt;
                    best.normal[0] = n[0]; best.normal[1] = n[1]; best.normal[2] = n[2];
                    best.mesh = (int)mi;
                }
            }
        }
    }
    return best;
}

}  // namespace pdrb
