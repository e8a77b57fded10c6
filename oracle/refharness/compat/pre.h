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
