// ORACLE / TEST INFRASTRUCTURE.  Probe record: one vector<double> per recorded tick, filled in the
// same order by the reference-TU harness and by the CPU restatement; a parallel vector of names is
// written beside the data so a test can assert both sides agree on the layout.
#pragma once
#include <vector>
#include <string>
#include <cstdio>
#include <cstdint>

namespace pdoracle {

struct Probe {
    std::vector<double> v;
    std::vector<std::string>* names = nullptr;  // filled on the first record only
    void p(const char* name, double x) {
        if (names) names->push_back(name);
        v.push_back(x);
    }
    void p3(const char* name, const float* x) {
        char b[96];
        for (int i = 0; i < 3; ++i) { snprintf(b, sizeof(b), "%s.%c", name, "xyz"[i]); p(b, x[i]); }
    }
    void pn(const char* name, const float* x, int n) {
        char b[96];
        for (int i = 0; i < n; ++i) { snprintf(b, sizeof(b), "%s[%d]", name, i); p(b, x[i]); }
    }
};

struct ProbeFile {
    FILE* f = nullptr;
    std::vector<std::string> names;
    std::vector<int32_t> ticks;
    std::vector<float> actions;
    std::vector<double> data;
    int nfields = 0;
    void add(int tick, float a0, float a1, const Probe& pr) {
        if (nfields == 0) nfields = (int)pr.v.size();
        ticks.push_back(tick);
        actions.push_back(a0);
        actions.push_back(a1);
        data.insert(data.end(), pr.v.begin(), pr.v.end());
    }
    // layout: int32 magic, nfields, nrec; names as '\n'-joined text (int32 length + bytes);
    // int32 ticks[nrec]; float32 actions[nrec][2]; float64 data[nrec][nfields]
    bool write(const char* path) const {
        FILE* fp = fopen(path, "wb");
        if (!fp) return false;
        const int32_t magic = 0x50444F52;  // 'PDOR'
        const int32_t nf = nfields, nr = (int32_t)ticks.size();
        fwrite(&magic, 4, 1, fp); fwrite(&nf, 4, 1, fp); fwrite(&nr, 4, 1, fp);
        std::string all;
        for (size_t i = 0; i < names.size(); ++i) { all += names[i]; all += '\n'; }
        const int32_t nl = (int32_t)all.size();
        fwrite(&nl, 4, 1, fp); fwrite(all.data(), 1, all.size(), fp);
        fwrite(ticks.data(), 4, ticks.size(), fp);
        fwrite(actions.data(), 4, actions.size(), fp);
        fwrite(data.data(), 8, data.size(), fp);
        fclose(fp);
        return true;
    }
};

}  // namespace pdoracle
