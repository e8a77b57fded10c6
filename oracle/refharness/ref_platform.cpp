// ORACLE / TEST INFRASTRUCTURE (fixture generator; runs only in the build container).
//
// POSIX implementation of the reference's platform interfaces Core/OS.h, Core/String.h and
// Core/Diag.h.  The reference's own Core/OS.cpp is Win32-only (OS.cpp:246-248 `#error`), and
// Core/String.cpp / Core/Diag.cpp rely on MSVC's wide printf convention (%s = wide string), so
// these three translation units are replaced here.  They are I/O plumbing (SURVEY.md §2 row 7,
// out of scope) and carry no simulation arithmetic.
#include "Core/OS.h"
#include "Core/String.h"
#include "Core/Diag.h"
#include <sys/stat.h>
#include <unistd.h>
#include <cwchar>
#include <algorithm>

namespace D {

static bool g_logEnabled = false;
void ref_enable_log(bool v) { g_logEnabled = v; }

// ---- String.h ----
std::string stra(const wchar_t* str) { std::wstring w(str); return std::string(w.begin(), w.end()); }
std::string stra(const std::wstring& w) { return std::string(w.begin(), w.end()); }
std::wstring strw(const char* str) { std::string s(str); return std::wstring(s.begin(), s.end()); }
std::wstring strw(const std::string& s) { return std::wstring(s.begin(), s.end()); }

std::string strafv(const char* format, va_list args) {
    char buf[1024];
    int n = vsnprintf(buf, sizeof(buf), format, args);
    std::string s;
    if (n > 0) s.assign(buf, std::min<size_t>((size_t)n, sizeof(buf) - 1));
    return s;
}
std::string straf(const char* format, ...) {
    va_list a; va_start(a, format);
    auto s = strafv(format, a);
    va_end(a);
    return s;
}
// MSVC wide printf: %s = wchar_t*, %S = char*.  glibc: %ls / %s.
static std::wstring msvc_to_glibc(const wchar_t* f) {
    std::wstring o;
    for (const wchar_t* p = f; *p; ++p) {
        if (*p == L'%') {
            o.push_back(*p++);
            while (*p && wcschr(L"-+ #0123456789.*", *p)) o.push_back(*p++);
            if (*p == L's') { o += L"ls"; continue; }
            if (*p == L'S') { o += L"s"; continue; }
            if (!*p) break;
            o.push_back(*p);
        } else o.push_back(*p);
    }
    return o;
}
std::wstring strwfv(const wchar_t* format, va_list args) {
    wchar_t buf[1024];
    std::wstring f = msvc_to_glibc(format);
    int n = vswprintf(buf, 1024, f.c_str(), args);
    std::wstring s;
    if (n > 0) s.assign(buf, n);
    return s;
}
std::wstring strwf(const wchar_t* format, ...) {
    va_list a; va_start(a, format);
    auto s = strwfv(format, a);
    va_end(a);
    return s;
}
template <typename T>
static std::vector<T> tsplit(const T& s, const T& delim) {
    std::vector<T> res;
    size_t start = 0;
    for (;;) {
        size_t end = s.find(delim, start);
        if (end != T::npos) { res.emplace_back(s.substr(start, end - start)); start = end + delim.length(); }
        else { res.emplace_back(s.substr(start)); break; }
    }
    return res;
}
std::vector<std::wstring> split(const std::wstring& s, const std::wstring& d) { return tsplit(s, d); }
std::vector<std::string> split(const std::string& s, const std::string& d) { return tsplit(s, d); }
void replace(std::wstring& s, wchar_t from, wchar_t to) { std::replace(s.begin(), s.end(), from, to); }
void replace(std::wstring& s, const std::wstring& from, const std::wstring& to) {
    size_t p = 0;
    while ((p = s.find(from, p)) != std::wstring::npos) { s.replace(p, from.length(), to); p += to.length(); }
}
bool ends_with(const std::wstring& s, wchar_t ch) { return !s.empty() && s.back() == ch; }

// ---- Diag.h ----
void log_set_file(const wchar_t*, bool) {}
void log_clear_file() {}
void log_printf(const wchar_t* format, ...) {
    if (!g_logEnabled) return;
    va_list a; va_start(a, format);
    auto s = strwfv(format, a);
    va_end(a);
    fprintf(stderr, "[ref] %s\n", stra(s).c_str());
}
void trace_warn(const wchar_t* msg, const char* file, int line) {
    if (g_logEnabled) fprintf(stderr, "[ref] WARN %s %s:%d\n", stra(msg).c_str(), file, line);
}
void trace_error(const wchar_t* msg, const char* file, int line) {
    fprintf(stderr, "[ref] ERROR %s %s:%d\n", stra(msg).c_str(), file, line);
}

// ---- OS.h ----
void osTraceDebug(const wchar_t*) {}
unsigned int osGetCurrentProcessId() { return (unsigned)getpid(); }
unsigned int osGetCurrentThreadId() { return 1; }
unsigned int osGetCurrentTicks() { return 0; }
void* osLoadLibraryA(const char*) { return nullptr; }
void* osLoadLibraryW(const wchar_t*) { return nullptr; }
void* osGetProcAddress(void*, const char*) { return nullptr; }
std::wstring osGetModuleFullPath() { return L""; }
std::wstring osGetCurrentDir() { return L""; }
void osSetCurrentDir(const std::wstring&) {}
std::wstring osCanonicPath(const std::wstring& p) { return p; }
std::wstring osCombinePath(const std::wstring& a, const std::wstring& b) { return a + L"/" + b; }
std::wstring osGetDirPath(const std::wstring& path) {
    for (int i = (int)path.length() - 1; i >= 0; --i)
        if (path[i] == L'\\' || path[i] == L'/') return path.substr(0, (size_t)i + 1);
    return std::wstring();
}
std::wstring osGetFileName(const std::wstring& path) {
    for (int i = (int)path.length() - 1; i >= 0; --i)
        if (path[i] == L'\\' || path[i] == L'/') return path.substr((size_t)i + 1);
    return std::wstring();
}
bool osFileExists(const std::wstring& path) {
    struct stat st;
    return stat(stra(path).c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
bool osDirExists(const std::wstring& path) {
    struct stat st;
    return stat(stra(path).c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
void osEnsureDirExists(const std::wstring& path) { if (!osDirExists(path)) mkdir(stra(path).c_str(), 0755); }
void osCreateDirectoryTree(const std::wstring& path) {
    std::wstring::size_type pos = 0;
    do { pos = path.find_first_of(L"\\/", pos + 1); osEnsureDirExists(path.substr(0, pos)); } while (pos != std::wstring::npos);
}
void* osFindWindow(const wchar_t*, const wchar_t*) { return nullptr; }
void* osFindProcessWindow(unsigned int) { return nullptr; }

bool FileHandle::open(const wchar_t* filename, const wchar_t* mode) {
    close();
    fd = fopen(stra(filename).c_str(), stra(mode).c_str());
    return fd != nullptr;
}
void FileHandle::close() { if (fd) { fclose(fd); fd = nullptr; } }
size_t FileHandle::size() const {
    if (!fd) return 0;
    fseek(fd, 0, SEEK_END);
    const size_t s = (size_t)ftell(fd);
    fseek(fd, 0, SEEK_SET);
    return s;
}

}  // namespace D

// ---- Core/SharedMemory.h (Win32 file mapping; interop is disabled: cfg/sim.ini [INTEROP] ENABLED=0) ----
#include <cstddef>
#include "Core/SharedMemory.h"
namespace D {
SharedMemory::SharedMemory() {}
SharedMemory::~SharedMemory() {}
void SharedMemory::allocate(const wchar_t*, size_t) {}
}

// MSVC C runtime rand() / srand() (see compat/pre.h)
static unsigned int g_holdrand = 1;
extern "C" int ref_msvc_rand(void) { g_holdrand = g_holdrand * 214013u + 2531011u; return (int)((g_holdrand >> 16) & 0x7fffu); }
extern "C" void ref_msvc_srand(unsigned int seed) { g_holdrand = seed; }
