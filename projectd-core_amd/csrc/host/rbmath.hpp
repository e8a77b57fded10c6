// pdbatch host side: rigid-body pose helpers used at model-build and reset time (the per-tick
// arithmetic lives in the HIP kernels).  Semantics follow what the reference obtains from ODE 0.16.3
// through Physics/ODE/RigidBodyODE.cpp:64-180 (mass box, dBodySetRotation, frame transforms) and
// Physics/ODE/JointODE.cpp:21-89 (anchor / axis setters taking world coordinates).
#pragma once
#include <cmath>
#include <cstring>
#include "pdb_types.h"

// the pose helpers and the reset edits (reset_core.hpp) are compiled for the host library (g++) and, unchanged, for the HIP
// kernels (hipcc): IEEE single precision, contraction off on both sides, so both produce the same bits
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PDB_HD __host__ __device__ __attribute__((always_inline))
#else
#define PDB_HD
#endif

namespace pdb {

struct HBody {
    float pos[3] = {0, 0, 0};
    float q[4] = {1, 0, 0, 0};
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float mass = 1, inertia[3] = {1, 1, 1};
};

PDB_HD inline void hCross(float* r, const float* a, const float* b) {
    const float r0 = a[1] * b[2] - a[2] * b[1], r1 = a[2] * b[0] - a[0] * b[2], r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
PDB_HD inline void hMul0(float* r, const float* M, const float* v) {  // M v
    const float r0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    const float r1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    const float r2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
PDB_HD inline void hMul1(float* r, const float* M, const float* v) {  // M^T v
    const float r0 = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    const float r1 = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    const float r2 = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
PDB_HD inline void hNorm3(float* v) {
    const float l = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (l > 0.0f) { const float s = 1.0f / sqrtf(l); v[0] *= s; v[1] *= s; v[2] *= s; }
    else { v[0] = 1; v[1] = 0; v[2] = 0; }
}
PDB_HD inline void hNorm4(float* q) {
    const float l = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (l > 0.0f) { const float s = 1.0f / sqrtf(l); q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
    else { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; }
}
PDB_HD inline void hQFromR(float q[4], const float R[9]) {
#define RR(i, j) R[(i) * 3 + (j)]
    const float tr = RR(0, 0) + RR(1, 1) + RR(2, 2);
    float s;
    if (tr >= 0) {
        s = sqrtf(tr + 1);
        q[0] = 0.5f * s;
        s = 0.5f * (1.0f / s);
        q[1] = (RR(2, 1) - RR(1, 2)) * s; q[2] = (RR(0, 2) - RR(2, 0)) * s; q[3] = (RR(1, 0) - RR(0, 1)) * s;
        return;
    }
    int c;
    if (RR(1, 1) > RR(0, 0)) c = (RR(2, 2) > RR(1, 1)) ? 2 : 1;
    else c = (RR(2, 2) > RR(0, 0)) ? 2 : 0;
    if (c == 0) {
        s = sqrtf((RR(0, 0) - (RR(1, 1) + RR(2, 2))) + 1);
        q[1] = 0.5f * s; s = 0.5f * (1.0f / s);
        q[2] = (RR(0, 1) + RR(1, 0)) * s; q[3] = (RR(2, 0) + RR(0, 2)) * s; q[0] = (RR(2, 1) - RR(1, 2)) * s;
    } else if (c == 1) {
        s = sqrtf((RR(1, 1) - (RR(2, 2) + RR(0, 0))) + 1);
        q[2] = 0.5f * s; s = 0.5f * (1.0f / s);
        q[3] = (RR(1, 2) + RR(2, 1)) * s; q[1] = (RR(0, 1) + RR(1, 0)) * s; q[0] = (RR(0, 2) - RR(2, 0)) * s;
    } else {
        s = sqrtf((RR(2, 2) - (RR(0, 0) + RR(1, 1))) + 1);
        q[3] = 0.5f * s; s = 0.5f * (1.0f / s);
        q[1] = (RR(2, 0) + RR(0, 2)) * s; q[2] = (RR(1, 2) + RR(2, 1)) * s; q[0] = (RR(1, 0) - RR(0, 1)) * s;
    }
#undef RR
}
// inv(qb) * qc
PDB_HD inline void hQMul1(float* qa, const float* qb, const float* qc) {
    const float a0 = qb[0] * qc[0] + qb[1] * qc[1] + qb[2] * qc[2] + qb[3] * qc[3];
    const float a1 = qb[0] * qc[1] - qb[1] * qc[0] - qb[2] * qc[3] + qb[3] * qc[2];
    const float a2 = qb[0] * qc[2] - qb[2] * qc[0] - qb[3] * qc[1] + qb[1] * qc[3];
    const float a3 = qb[0] * qc[3] - qb[3] * qc[0] - qb[1] * qc[2] + qb[2] * qc[1];
    qa[0] = a0; qa[1] = a1; qa[2] = a2; qa[3] = a3;
}
PDB_HD inline void hSetRotation(HBody& b, const float Rin[9]) {
    float m[9];
    memcpy(m, Rin, sizeof(m));
    const float n0 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    if (n0 != 1.0f) hNorm3(m);
    const float proj = m[0] * m[3] + m[1] * m[4] + m[2] * m[5];
    if (proj != 0.0f) { m[3] -= proj * m[0]; m[4] -= proj * m[1]; m[5] -= proj * m[2]; }
    const float n1 = m[3] * m[3] + m[4] * m[4] + m[5] * m[5];
    if (n1 != 1.0f) hNorm3(m + 3);
    hCross(m + 6, m, m + 3);
    memcpy(b.R, m, sizeof(m));
    hQFromR(b.q, Rin);
    hNorm4(b.q);
}
// mat44f (row-vector convention, rows = body axes) <-> R; RigidBodyODE.cpp:140-180
PDB_HD inline void hSetRotationM(HBody& b, const float M[9] /* M11..M33 */) {
    const float R[9] = {M[0], M[3], M[6], M[1], M[4], M[7], M[2], M[5], M[8]};
    hSetRotation(b, R);
}
PDB_HD inline void hWorldMatrix3(const HBody& b, float M[9]) {
    M[0] = b.R[0]; M[1] = b.R[3]; M[2] = b.R[6];
    M[3] = b.R[1]; M[4] = b.R[4]; M[5] = b.R[7];
    M[6] = b.R[2]; M[7] = b.R[5]; M[8] = b.R[8];
}
PDB_HD inline void hLocalToWorld(const HBody& b, const float* p, float* o) {
    float t[3];
    hMul0(t, b.R, p);
    o[0] = t[0] + b.pos[0]; o[1] = t[1] + b.pos[1]; o[2] = t[2] + b.pos[2];
}
PDB_HD inline void hWorldToLocal(const HBody& b, const float* p, float* o) {
    const float d[3] = {p[0] - b.pos[0], p[1] - b.pos[1], p[2] - b.pos[2]};
    hMul1(o, b.R, d);
}
PDB_HD inline void hBoxInertia(float m, float lx, float ly, float lz, float out[3]) {
    const float M = lx * ly * lz * 1.0f;
    float i0 = M / 12.0f * (ly * ly + lz * lz);
    float i1 = M / 12.0f * (lx * lx + lz * lz);
    float i2 = M / 12.0f * (lx * lx + ly * ly);
    const float scale = m / M;
    out[0] = i0 * scale; out[1] = i1 * scale; out[2] = i2 * scale;
}
// reference Core/Math.cpp:87-115 (mat44f::createFromAxisAngle), 3x3 part in M11..M33 order
PDB_HD inline void hAxisAngle(const float ax[3], float angle, float M[9]) {
    const float s = sinf(angle), c = cosf(angle), omc = 1.0f - c;
    M[0] = ((ax[0] * ax[0]) * omc) + c;
    M[4] = ((ax[1] * ax[1]) * omc) + c;
    M[8] = ((ax[2] * ax[2]) * omc) + c;
    M[1] = (ax[2] * s) + (ax[1] * ax[0]) * omc;
    M[5] = (ax[0] * s) + (ax[2] * ax[1]) * omc;
    M[6] = (ax[1] * s) + (ax[2] * ax[0]) * omc;
    M[2] = (ax[2] * ax[0]) * omc - (ax[1] * s);
    M[3] = (ax[1] * ax[0]) * omc - (ax[2] * s);
    M[7] = (ax[2] * ax[1]) * omc - (ax[0] * s);
}

}  // namespace pdb
