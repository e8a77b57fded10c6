// ORACLE / TEST INFRASTRUCTURE.  Elementary-function selection for the CPU restatement:
//   default                    glibc (what the reference's own build calls) -> liboracle.so, pinned against tests/golden
//   -DCPUREF_PORTABLE_MATH     the product's reproducible-math specification -> liboracle_pm.so, bit-comparable with the GPU
#pragma once
#include <cmath>
#ifdef CPUREF_PORTABLE_MATH
#include "pm_ref.h"
static inline float m_sinf(float x) { return pmref::r_sinf(x); }
static inline float m_cosf(float x) { return pmref::r_cosf(x); }
static inline float m_tanf(float x) { return pmref::r_tanf(x); }
static inline float m_atanf(float x) { return pmref::r_atanf(x); }
static inline float m_atan2f(float y, float x) { return pmref::r_atan2f(y, x); }
static inline float m_asinf(float x) { return pmref::r_asinf(x); }
static inline float m_acosf(float x) { return pmref::r_acosf(x); }
static inline float m_powf(float x, float y) { return pmref::r_powf(x, y); }
#else
static inline float m_sinf(float x) { return sinf(x); }
static inline float m_cosf(float x) { return cosf(x); }
static inline float m_tanf(float x) { return tanf(x); }
static inline float m_atanf(float x) { return atanf(x); }
static inline float m_atan2f(float y, float x) { return atan2f(y, x); }
static inline float m_asinf(float x) { return asinf(x); }
static inline float m_acosf(float x) { return acosf(x); }
static inline float m_powf(float x, float y) { return powf(x, y); }
#endif
