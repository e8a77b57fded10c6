// ORACLE / TEST INFRASTRUCTURE.  Scripted action sequences shared by the reference-TU harness
// (oracle/refharness) and the CPU restatement (oracle/cpu_ref).  Actions are the env's 2-vector
// (pyprojectd/projectd_env.py:157-160): a0 = steer in [-1,1], a1 -> gas = linscale(a1,-1,1,0.1,1).
#pragma once
#include <cmath>

namespace pdoracle {

struct Scenario { const char* name; int ticks; int denseTicks; int stride; };

static const Scenario kScenarios[] = {
    {"idle", 600, 200, 10},
    {"launch", 2000, 450, 10},
    {"circle", 1600, 300, 10},
    {"slalom", 2400, 300, 10},
};
static const int kNumScenarios = 4;

inline void scenarioAction(int sid, int tick, float& a0, float& a1) {
    const double t = (double)tick * (1.0 / 333.0);
    switch (sid) {
    case 0: a0 = 0.0f; a1 = -1.0f; break;
    case 1: a0 = 0.0f; a1 = 1.0f; break;
    case 2: a0 = 0.35f; a1 = 0.2f; break;
    default:
        a0 = (float)(0.4 * sin(6.283185307179586 * t / 2.0));
        a1 = (float)(0.6 * sin(6.283185307179586 * t / 5.0 + 1.0));
        break;
    }
}

inline bool scenarioRecord(const Scenario& s, int tick) {
    return tick < s.denseTicks || (tick % s.stride) == 0 || tick == s.ticks - 1;
}

// gas mapping, python double arithmetic then stored to a float32 field (utils_d.py:9-11)
inline float envGas(float a1) {
    double x = (double)a1;
    if (x < -1.0) x = -1.0;
    if (x > 1.0) x = 1.0;
    return (float)(((1.0 - 0.1) * (x - (-1.0))) / (1.0 - (-1.0)) + 0.1);
}

}  // namespace pdoracle
