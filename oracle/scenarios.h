// ORACLE / TEST INFRASTRUCTURE.  Scripted action sequences shared by the reference-TU harness
// (oracle/refharness) and the CPU restatement (oracle/cpu_ref).  Actions are the env's 2-vector
// (pyprojectd/projectd_env.py:157-160): a0 = steer in [-1,1], a1 -> gas = linscale(a1,-1,1,0.1,1).
#pragma once
#include <cmath>

namespace pdoracle {

// full: 0 = env 2-vector (steer, a1) with the env's assists (all on); 1 = every CarControls field scripted
// (PyProjectD.cpp:297-305 setCarControls) with the scenario's own assist switches (setCarAssists :307-317)
struct Scenario { const char* name; int ticks; int denseTicks; int stride; int full; int autoClutch, autoShift, autoBlip; };

static const Scenario kScenarios[] = {
    {"idle", 600, 200, 10, 0, 1, 1, 1},
    {"launch", 2000, 450, 10, 0, 1, 1, 1},
    {"circle", 1600, 300, 10, 0, 1, 1, 1},
    {"slalom", 2400, 300, 10, 0, 1, 1, 1},
    {"brake", 2400, 200, 10, 1, 1, 1, 1},    // pedal brake to a stop, handbrake turn (BrakeSystem, tyre lock)
    {"manual", 2600, 300, 10, 1, 0, 0, 0},   // no assists: manual clutch, gearUp/gearDn pulses, H-shifter gear select, grinding
};
static const int kNumScenarios = 6;

struct Ctl { float steer, clutch, brake, handBrake, gas; int requestedGearIndex, gearUp, gearDn; };

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

// every CarControls field for one tick; scenarios 0-3 reproduce the env mapping (other fields stay at their defaults)
inline float envGas(float a1);
inline void scenarioControls(int sid, int tick, Ctl& c) {
    c.steer = 0; c.clutch = 0; c.brake = 0; c.handBrake = 0; c.gas = 0; c.requestedGearIndex = -1; c.gearUp = 0; c.gearDn = 0;
    const double t = (double)tick * (1.0 / 333.0);
    if (sid < 4) { float a0, a1; scenarioAction(sid, tick, a0, a1); c.steer = a0; c.gas = envGas(a1); return; }
    if (sid == 4) {
        if (t < 3.0) { c.gas = 1.0f; }
        else if (t < 5.0) { c.brake = 0.8f; }
        else if (t < 6.0) { c.gas = 0.7f; c.steer = 0.3f; }
        else if (t < 6.6) { c.handBrake = 1.0f; c.steer = 0.5f; c.gas = 0.2f; }
        else { c.gas = 0.5f; c.steer = -0.2f; c.brake = (float)(0.15 * (1.0 + sin(6.283185307179586 * t))); }
        return;
    }
    // manual: clutch 1 = engaged (env sets clutch=1 when auto_clutch is off, projectd_env.py:163-164)
    if (t < 0.2) { c.clutch = 0.0f; c.gas = 0.3f; c.gearUp = (tick >= 20 && tick < 26) ? 1 : 0; }
    else if (t < 1.5) { c.clutch = (float)((t - 0.2) / 1.3); c.gas = 0.6f; }
    else if (t < 3.0) { c.clutch = 1.0f; c.gas = 0.9f; }
    else if (t < 3.3) { c.clutch = 0.0f; c.gas = 0.0f; c.gearUp = (t >= 3.05 && t < 3.08) ? 1 : 0; }
    else if (t < 5.0) { c.clutch = 1.0f; c.gas = 0.8f; }
    else if (t < 5.3) { c.clutch = 0.2f; c.gas = 0.1f; c.gearDn = (t >= 5.05 && t < 5.08) ? 1 : 0; }
    else if (t < 6.0) { c.clutch = 1.0f; c.gas = 0.3f; c.brake = 0.3f; }
    else if (t < 6.5) { c.clutch = 1.0f; c.gas = 0.4f; c.requestedGearIndex = 4; }      // select 3rd with the clutch engaged: grinds
    else if (t < 6.7) { c.clutch = 0.0f; c.gas = 0.0f; c.requestedGearIndex = 4; }      // clutch in: the gear goes in
    else { c.clutch = 1.0f; c.gas = 0.7f; c.requestedGearIndex = 4; c.steer = 0.15f; }
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
