// pdbatch device side (gfx950): small vector / LUT helpers.  Arithmetic is IEEE single precision with
// contraction disabled (-ffp-contract=off) so that +,-,*,/ and sqrt evaluate exactly like the scalar
// CPU restatement; explicit fmaf is used only inside the LDL^T solver.
#pragma once
#include <hip/hip_runtime.h>
#include "pdb_types.h"

#define PDB_DEV __device__ __forceinline__

struct DV3 {
    float x, y, z;
};
PDB_DEV DV3 mk3(float x, float y, float z) { DV3 r; r.x = x; r.y = y; r.z = z; return r; }
PDB_DEV DV3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
PDB_DEV void st3(float* p, const DV3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
PDB_DEV DV3 operator+(const DV3& a, const DV3& b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PDB_DEV DV3 operator-(const DV3& a, const DV3& b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PDB_DEV DV3 operator*(const DV3& a, float f) { return mk3(a.x * f, a.y * f, a.z * f); }
PDB_DEV DV3 operator/(const DV3& a, float f) { return mk3(a.x / f, a.y / f, a.z / f); }
PDB_DEV float dot(const DV3& a, const DV3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PDB_DEV DV3 cross(const DV3& a, const DV3& b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PDB_DEV float sqlen(const DV3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PDB_DEV float len3(const DV3& a) { return sqrtf(sqlen(a)); }
// vec3f::norm(l): scale by 1/l when l != 0 (reference Core/Math.h:121)
PDB_DEV DV3 normL(const DV3& a, float l) { if (l != 0.0f) { const float s = 1.0f / l; return mk3(a.x * s, a.y * s, a.z * s); } return a; }
PDB_DEV DV3 norm3(const DV3& a) { return normL(a, len3(a)); }

PDB_DEV float fsign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
PDB_DEV float fclamp(float x, float a, float b) { return x < a ? a : (x > b ? b : x); }
PDB_DEV float fmaxr(float a, float b) { return a > b ? a : b; }   // tmax(a,b) = a > b ? a : b
PDB_DEV float fminr(float a, float b) { return a < b ? a : b; }   // tmin(a,b) = a < b ? a : b
PDB_DEV float linscale(float x, float x0, float x1, float r0, float r1) {
    x = fclamp(x, x0, x1);
    return ((r1 - r0) * (x - x0)) / (x1 - x0) + r0;
}

// 3x3 row-major helpers
PDB_DEV DV3 mulM(const float* M, const DV3& v) {   // M v
    return mk3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z);
}
PDB_DEV DV3 mulMT(const float* M, const DV3& v) {  // M^T v
    return mk3(M[0] * v.x + M[3] * v.y + M[6] * v.z, M[1] * v.x + M[4] * v.y + M[7] * v.z, M[2] * v.x + M[5] * v.y + M[8] * v.z);
}

// clamp-ended piecewise-linear LUT (reference Core/Curve.cpp:94-115): value at the first knot i >= 1 with ref <= x[i].
// The knots are non-decreasing (the loader refuses curves that are not), so that knot is found by counting the knots
// below ref -- 24 independent compares instead of a data-dependent loop of dependent loads -- and one interpolation follows.
PDB_DEV float lut(const pdb_curve& c, float ref) {
    const int n = c.n;
    if (n == 0) return 0.0f;
    if (!(ref == ref)) return c.y[n - 1];   // NaN compares false everywhere in the reference's scan
    int below = 0;
#pragma unroll
    for (int i = 0; i < PDB_MAX_CURVE; ++i) below += (i < n && c.x[i] < ref) ? 1 : 0;
    if (below == 0) return c.y[0];          // ref <= x[0]
    if (below >= n) return c.y[n - 1];      // beyond the last knot
    const int i = below;                    // first knot with ref <= x[i], i >= 1
    const float x0 = c.x[i - 1], x1 = c.x[i], y0 = c.y[i - 1], y1 = c.y[i];
    return (((y1 - y0) * (ref - x0)) / (x1 - x0)) + y0;
}
