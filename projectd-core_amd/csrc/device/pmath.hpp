// pdbatch: reproducible elementary functions for the per-tick path (host + device).
//
// The vehicle step is full of knife-edge logic (sign flips of near-zero wheel speeds, lock/unlock,
// gear thresholds), so a 1-ulp difference between two libm implementations grows into a different
// trajectory within tens of ticks.  To make the GPU path reproducible against a CPU evaluation, every
// transcendental on the hot path is evaluated here from IEEE double +,-,*,/ and sqrt only (all
// correctly rounded on gfx950 and x86-64, contraction disabled), then rounded once to float:
//   sin/cos : Cody-Waite reduction by pi/2 (two-part constant), Taylor kernels on [-pi/4, pi/4]
//   atan    : two half-angle steps  atan x = 2 atan( x / (1 + sqrt(1 + x^2)) ), odd Taylor series
//   asin/acos/atan2 : via atan and sqrt((1-x)(1+x))
//   pow     : exp( y * log x ), log by m*2^e split and the atanh series, exp by ln2 reduction + Taylor
// Errors are ~1e-15 relative before the final rounding, i.e. the float result is the correctly rounded
// value except for rare near-ties.  Specification = this sequence of operations; the test oracle
// implements the same specification independently (oracle/cpu_ref/pm_ref.h).
#pragma once
#include <stdint.h>
#include <string.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define PM_FN __host__ __device__ inline
#else
#define PM_FN inline
#endif

namespace pm {

PM_FN double pm_floor(double x) {
    // exact floor for |x| < 2^51 without calling libm
    const double t = (double)(long long)x;
    return (t > x) ? t - 1.0 : t;
}
PM_FN double pm_sqrt(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __dsqrt_rn(x);
#else
    return __builtin_sqrt(x);
#endif
}
PM_FN double pm_fabs(double x) { return x < 0.0 ? -x : x; }

PM_FN void sincos_k(double x, double& s, double& c) {
    const double k = pm_floor(x * 0.63661977236758134308 + 0.5);
    // pi/2 split: hi has 33 significant bits so k*hi is exact for |k| < 2^20
    const double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = r * r;
    // Taylor kernels in Horner form; the reciprocal factorials are compile-time constants (no run-time division)
    const double sp = r + r * (z * (-1.0 / 6.0 + z * (1.0 / 120.0 + z * (-1.0 / 5040.0 + z * (1.0 / 362880.0 + z * (-1.0 / 39916800.0 + z * (1.0 / 6227020800.0 + z * (-1.0 / 1307674368000.0))))))));
    const double cp = 1.0 + z * (-1.0 / 2.0 + z * (1.0 / 24.0 + z * (-1.0 / 720.0 + z * (1.0 / 40320.0 + z * (-1.0 / 3628800.0 + z * (1.0 / 479001600.0 + z * (-1.0 / 87178291200.0 + z * (1.0 / 20922789888000.0))))))));
    const long long q = ((long long)k) & 3;
    if (q == 0) { s = sp; c = cp; }
    else if (q == 1) { s = cp; c = -sp; }
    else if (q == 2) { s = -sp; c = -cp; }
    else { s = -cp; c = sp; }
}
PM_FN float sinf_(float x) { double s, c; sincos_k((double)x, s, c); return (float)s; }
PM_FN float cosf_(float x) { double s, c; sincos_k((double)x, s, c); return (float)c; }
PM_FN float tanf_(float x) { double s, c; sincos_k((double)x, s, c); return (float)(s / c); }

PM_FN double atan_k(double x) {
    const bool neg = x < 0.0;
    double a = neg ? -x : x;
    const bool inv = a > 1.0;
    if (inv) a = 1.0 / a;
    a = a / (1.0 + pm_sqrt(1.0 + a * a));
    a = a / (1.0 + pm_sqrt(1.0 + a * a));   // a <= tan(pi/16) = 0.1989
    const double z = a * a;
    // atan a = a (1 - z/3 + z^2/5 - ... ) up to z^13/27
    double p = 1.0 / 27.0;
    p = 1.0 / 25.0 - z * p; p = 1.0 / 23.0 - z * p; p = 1.0 / 21.0 - z * p; p = 1.0 / 19.0 - z * p; p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p; p = 1.0 / 13.0 - z * p; p = 1.0 / 11.0 - z * p; p = 1.0 / 9.0 - z * p; p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p; p = 1.0 / 3.0 - z * p; p = 1.0 - z * p;
    double r = 4.0 * (a * p);
    if (inv) r = 1.57079632679489661923 - r;
    return neg ? -r : r;
}
PM_FN double atan2_k(double y, double x) {
    if (x > 0.0) return atan_k(y / x);
    if (x < 0.0) return (y >= 0.0) ? atan_k(y / x) + 3.14159265358979323846 : atan_k(y / x) - 3.14159265358979323846;
    if (y > 0.0) return 1.57079632679489661923;
    if (y < 0.0) return -1.57079632679489661923;
    return 0.0;
}
PM_FN float atanf_(float x) { return (float)atan_k((double)x); }
PM_FN float atan2f_(float y, float x) { return (float)atan2_k((double)y, (double)x); }
PM_FN float asinf_(float x) { const double d = (double)x; return (float)atan2_k(d, pm_sqrt((1.0 - d) * (1.0 + d))); }
PM_FN float acosf_(float x) { const double d = (double)x; return (float)atan2_k(pm_sqrt((1.0 - d) * (1.0 + d)), d); }

PM_FN double log_k(double x) {   // x > 0, normal
    uint64_t u; memcpy(&u, &x, 8);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m; memcpy(&m, &u, 8);
    if (m > 1.41421356237309504880) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 23.0;
    p = 1.0 / 21.0 + z * p; p = 1.0 / 19.0 + z * p; p = 1.0 / 17.0 + z * p; p = 1.0 / 15.0 + z * p; p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p; p = 1.0 / 9.0 + z * p; p = 1.0 / 7.0 + z * p; p = 1.0 / 5.0 + z * p; p = 1.0 / 3.0 + z * p; p = 1.0 + z * p;
    return (double)e * 0.69314718055994530942 + 2.0 * (s * p);
}
PM_FN double exp_k(double z) {
    if (z > 700.0) z = 700.0;
    if (z < -700.0) return 0.0;
    const double k = pm_floor(z * 1.44269504088896340736 + 0.5);
    const double r = (z - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    // e^r = sum r^n / n!, n <= 13, Horner with compile-time reciprocal factorials
    double p = 1.0 / 6227020800.0;
    p = 1.0 / 479001600.0 + r * p; p = 1.0 / 39916800.0 + r * p; p = 1.0 / 3628800.0 + r * p; p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p; p = 1.0 / 5040.0 + r * p; p = 1.0 / 720.0 + r * p; p = 1.0 / 120.0 + r * p; p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p; p = 0.5 + r * p; p = 1.0 + r * p; p = 1.0 + r * p;
    const uint64_t u = (uint64_t)((long long)k + 1023) << 52;
    double sc; memcpy(&sc, &u, 8);
    return p * sc;
}
// x^y for the hot path's uses (x >= 0); x == 0 -> 0 for y > 0, 1 for y == 0
PM_FN float powf_(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (!(x > 0.0f)) return 0.0f;
    if (x == 1.0f) return 1.0f;
    return (float)exp_k((double)y * log_k((double)x));
}

}  // namespace pm
