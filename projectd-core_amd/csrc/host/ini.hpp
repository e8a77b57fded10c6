// pdbatch host side: readers for the reference's on-disk text formats.
//   INI   : "[SECTION]" / "KEY=VALUE ; comment"   (reference Core/INIReader.cpp:31-107)
//   .lut  : "x|y" per line, ';' comments            (reference Core/Curve.cpp:128-175)
//   inline curve "(|0=0.0|4000=1.5|)"               (reference Core/Curve.cpp:177-200)
// Same acceptance rules as the reference reader: first occurrence of a key wins, values are not
// trimmed (numeric parsing skips leading blanks and stops at the first non-numeric character).
#pragma once
#include <map>
#include <string>
#include <vector>
#include <fstream>
#include <sstream>
#include <cstdlib>
#include <stdexcept>
#include "pdb_types.h"

namespace pdb {

inline std::vector<std::string> splitStr(const std::string& s, const std::string& delim) {
    std::vector<std::string> res;
    size_t start = 0;
    for (;;) {
        size_t end = s.find(delim, start);
        if (end != std::string::npos) { res.emplace_back(s.substr(start, end - start)); start = end + delim.size(); }
        else { res.emplace_back(s.substr(start)); break; }
    }
    return res;
}

inline float toFloat(const std::string& s) {
    const char* p = s.c_str();
    char* e = nullptr;
    float v = strtof(p, &e);
    if (e == p) throw std::runtime_error("pdb: bad float '" + s + "'");
    return v;
}
inline int toInt(const std::string& s) {
    const char* p = s.c_str();
    char* e = nullptr;
    long v = strtol(p, &e, 10);
    if (e == p) throw std::runtime_error("pdb: bad int '" + s + "'");
    return (int)v;
}

struct Ini {
    std::map<std::string, std::map<std::string, std::string>> sections;
    std::string dir;
    bool ready = false;

    Ini() {}
    explicit Ini(const std::string& path) { load(path); }

    bool load(const std::string& path) {
        sections.clear();
        ready = false;
        const size_t sl = path.find_last_of("/\\");
        dir = (sl == std::string::npos) ? std::string() : path.substr(0, sl + 1);
        std::ifstream fs(path);
        if (!fs.is_open()) return false;
        std::string line, secName;
        std::map<std::string, std::string>* sec = nullptr;
        while (std::getline(fs, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            const size_t comm = line.find(';');
            size_t s1 = line.find('[');
            if (s1 != std::string::npos) {
                s1++;
                const size_t s2 = line.find(']');
                if (s2 != std::string::npos && s2 > s1 && s2 < comm) {
                    secName = line.substr(s1, s2 - s1);
                    sec = &sections.insert({secName, {}}).first->second;
                }
            } else if (!secName.empty()) {
                const size_t split = line.find('=');
                if (split != std::string::npos && split > 0 && split < comm) {
                    size_t eol = line.find_first_of(";\n", split);
                    if (eol != std::string::npos) eol -= split;
                    sec->insert({line.substr(0, split), line.substr(split + 1, eol)});
                }
            }
        }
        ready = true;
        return true;
    }
    bool hasSection(const std::string& s) const { return sections.count(s) != 0; }
    bool hasKey(const std::string& s, const std::string& k) const {
        auto it = sections.find(s);
        return it != sections.end() && it->second.count(k) != 0;
    }
    std::string getString(const std::string& s, const std::string& k) const {
        auto it = sections.find(s);
        if (it != sections.end()) {
            auto jt = it->second.find(k);
            if (jt != it->second.end()) return jt->second;
        }
        return std::string();
    }
    int getInt(const std::string& s, const std::string& k) const {
        const std::string v = getString(s, k);
        return v.empty() ? 0 : toInt(v);
    }
    float getFloat(const std::string& s, const std::string& k) const {
        const std::string v = getString(s, k);
        return v.empty() ? 0.0f : toFloat(v);
    }
    void getFloat3(const std::string& s, const std::string& k, float out[3]) const {
        out[0] = out[1] = out[2] = 0;
        auto v = splitStr(getString(s, k), ",");
        if (v.size() == 3) { out[0] = toFloat(v[0]); out[1] = toFloat(v[1]); out[2] = toFloat(v[2]); }
    }
    bool tryGetFloat(const std::string& s, const std::string& k, float& out) const {
        if (!hasKey(s, k)) return false;
        out = getFloat(s, k);
        return true;
    }
    bool tryGetInt(const std::string& s, const std::string& k, int& out) const {
        if (!hasKey(s, k)) return false;
        out = getInt(s, k);
        return true;
    }
};

inline void curveAdd(pdb_curve& c, float x, float y) {
    if (c.n >= PDB_MAX_CURVE) throw std::runtime_error("pdb: curve has more than PDB_MAX_CURVE points");
    c.x[c.n] = x; c.y[c.n] = y; c.n++;
}
inline bool curveLoad(pdb_curve& c, const std::string& path) {
    c.n = 0;
    std::ifstream fs(path);
    if (!fs.is_open()) return false;
    std::string line;
    while (std::getline(fs, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const size_t comm = line.find(';');
        if (comm != std::string::npos) line = line.substr(0, comm);
        if (line.empty()) continue;
        auto kv = splitStr(line, "|");
        if (kv.size() == 2) curveAdd(c, toFloat(kv[0]), toFloat(kv[1]));
    }
    return c.n > 0;
}
inline bool curveParseInline(pdb_curve& c, const std::string& str) {
    c.n = 0;
    size_t p1 = str.find('('), p2 = str.find(')');
    if (p1 != std::string::npos && p2 != std::string::npos && p1 < p2) {
        p1++;
        for (auto& pair : splitStr(str.substr(p1, p2 - p1), "|")) {
            if (pair.empty()) continue;
            auto kv = splitStr(pair, "=");
            if (kv.size() == 2) curveAdd(c, toFloat(kv[0]), toFloat(kv[1]));
        }
    }
    return c.n > 0;
}
// clamp-ended linear LUT (reference Core/Curve.cpp:94-115)
inline float curveValue(const pdb_curve& c, float ref) {
    if (c.n == 0) return 0.0f;
    if (ref <= c.x[0]) return c.y[0];
    for (int i = 1; i < c.n; ++i)
        if (ref <= c.x[i]) return (((c.y[i] - c.y[i - 1]) * (ref - c.x[i - 1])) / (c.x[i] - c.x[i - 1])) + c.y[i - 1];
    return c.y[c.n - 1];
}

}  // namespace pdb
