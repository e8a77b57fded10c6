// Force-included prelude: standard headers MSVC pulls in transitively, and two MSVC CRT names.
#pragma once
#include <algorithm>
#include <cstring>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cwchar>
#include <string>
#include <locale>
#include <atomic>
using std::isfinite;
inline int _wfopen_s(FILE** fd, const wchar_t* n, const wchar_t* m) {
    std::wstring wn(n), wm(m);
    std::string sn(wn.begin(), wn.end()), sm(wm.begin(), wm.end());
    *fd = fopen(sn.c_str(), sm.c_str());
    return *fd ? 0 : 1;
}

// The reference's randR (Core/Math.h:49-52) is rand() / RAND_MAX of its C runtime: the reference is built with MSVC, whose
// generator is holdrand = holdrand * 214013 + 2531011, rand() = (holdrand >> 16) & 0x7fff, RAND_MAX = 0x7fff.  The harness
// gives the reference TUs that runtime's rand (ref_platform.cpp) instead of glibc's, so that Car::teleportByMode(Random)
// draws what the reference binary draws.
#undef RAND_MAX
#define RAND_MAX 0x7fff
extern "C" int ref_msvc_rand(void);
extern "C" void ref_msvc_srand(unsigned int seed);
namespace std { using ::ref_msvc_rand; using ::ref_msvc_srand; }
#define rand ref_msvc_rand
#define srand ref_msvc_srand
