// ORACLE / TEST INFRASTRUCTURE -- C API over cpu_ref for ctypes (tests, smoke, bench cpu_baseline leg).
#include "cpu_ref.h"
#include "../scenarios.h"
#include "pm_ref.h"
#include <cmath>
#include <cstring>
#include <vector>
#include <string>
#include <chrono>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

struct CpuRefHandle {
    pdb_car_params P;
    std::vector<uint8_t> blob;
    cpuref::TrackData T;
    pdb_dyn_state s0;
    cpuref::Car car;
    void (*hook)(pdb_dyn_state*, int) = nullptr;
};

extern "C" {

void* cpuref_create(const pdb_car_params* P, const void* trackBlob, uint64_t trackBytes, const pdb_dyn_state* s0) {
    auto* h = new CpuRefHandle();
    h->P = *P;
    h->blob.assign((const uint8_t*)trackBlob, (const uint8_t*)trackBlob + trackBytes);
    h->T.bind(h->blob.data());
    h->s0 = *s0;
    h->car.init(&h->P, &h->T, *s0);
    return h;
}
void cpuref_destroy(void* hh) { delete (CpuRefHandle*)hh; }
void cpuref_set_state(void* hh, const pdb_dyn_state* s) { ((CpuRefHandle*)hh)->car.loadState(*s); }
void cpuref_get_state(void* hh, pdb_dyn_state* s) { *s = ((CpuRefHandle*)hh)->car.S; }
// Car::teleportByMode for the in-tick auto-teleport (pdb_car_params.autoTeleport): the product's pdb_teleport_by_mode, handed in by the test
void cpuref_set_auto_teleport_hook(void* hh, void (*hook)(pdb_dyn_state*, int)) { ((CpuRefHandle*)hh)->car.autoTeleportHook = hook; ((CpuRefHandle*)hh)->hook = hook; }
// the contact joints alive in the engine's group: PDB_MAX_CONTACTS entries, the first S.numContacts meaningful
void cpuref_get_contacts(void* hh, pdb_contact* c) { ((CpuRefHandle*)hh)->car.getContacts(c); }
int cpuref_contact_candidates(void* hh) { return ((CpuRefHandle*)hh)->car.contactCandidates; }
void cpuref_set_contacts(void* hh, const pdb_contact* c, int n) { ((CpuRefHandle*)hh)->car.setContacts(c, n); }
// contact rows of the last tick's solve (3 per contact: normal, friction 1, friction 2): lambda, lo, hi; returns the row count
int cpuref_last_contact_rows(void* hh, float* lambda, float* lo, float* hi, int cap, int* iterations) {
    const pdrb::World& w = ((CpuRefHandle*)hh)->car.w;
    const int n = (int)w.lastContactLambda.size();
    for (int i = 0; i < n && i < cap; ++i) { lambda[i] = w.lastContactLambda[i]; lo[i] = w.lastContactLo[i]; hi[i] = w.lastContactHi[i]; }
    if (iterations) *iterations = w.lastLcpIterations;
    return n;
}
// multi-car simulators: a car's slipstream as its last tick left it; the OTHER cars' slipstreams its next tick reads (n of them, in the simulator's car order)
void cpuref_get_slip(void* hh, pdb_slip_state* out) { *out = ((CpuRefHandle*)hh)->car.slip; }
void cpuref_set_slip(void* hh, const pdb_slip_state* in) { ((CpuRefHandle*)hh)->car.slip = *in; }
void cpuref_set_guid(void* hh, int guid) { ((CpuRefHandle*)hh)->car.physicsGUID = guid; }
void cpuref_set_other_slips(void* hh, const pdb_slip_state* in, int n) { ((CpuRefHandle*)hh)->car.otherSlips.assign(in, in + n); }
// controls-level step: steer, gas
void cpuref_step(void* hh, float steer, float gas) {
    ((CpuRefHandle*)hh)->car.step(steer, gas, (float)(1.0 / 333.0), 1.0 / 333.0);
}
// env-level step: a0, a1 (projectd_env.py:157-160)
void cpuref_step_env(void* hh, float a0, float a1) {
    ((CpuRefHandle*)hh)->car.step(a0, pdoracle::envGas(a1), (float)(1.0 / 333.0), 1.0 / 333.0);
}
// full-controls step: 8 floats laid out like PDB_ACTION_FULL (include/pdbatch.h)
static pdb_controls ctlOf(const float* a) {
    pdb_controls c; memset(&c, 0, sizeof(c));
    c.steer = a[0]; c.clutch = a[1]; c.brake = a[2]; c.handBrake = a[3]; c.gas = a[4];
    c.isShifterSupported = 1; c.requestedGearIndex = (int8_t)(int)a[5]; c.gearUp = a[6] != 0.0f; c.gearDn = a[7] != 0.0f;
    return c;
}
void cpuref_step_controls(void* hh, const float* a) {
    ((CpuRefHandle*)hh)->car.stepControls(ctlOf(a), (float)(1.0 / 333.0), 1.0 / 333.0);
}
// the scripted controls of oracle/scenarios.h as PDB_ACTION_FULL rows (tests drive the GPU with the same script)
void cpuref_scenario_controls(int sid, int tick, float* a) {
    pdoracle::Ctl c; pdoracle::scenarioControls(sid, tick, c);
    a[0] = c.steer; a[1] = c.clutch; a[2] = c.brake; a[3] = c.handBrake; a[4] = c.gas; a[5] = (float)c.requestedGearIndex; a[6] = (float)c.gearUp; a[7] = (float)c.gearDn;
}
int cpuref_scenario_info(int sid, int* ticks, int* full, int* assists3) {   // assists3[0..2] clutch/shift/blip, [3] smooth steering
    if (sid < 0 || sid >= pdoracle::kNumScenarios) return -1;
    const auto& sc = pdoracle::kScenarios[sid];
    *ticks = sc.ticks; *full = sc.full; assists3[0] = sc.autoClutch; assists3[1] = sc.autoShift; assists3[2] = sc.autoBlip; assists3[3] = sc.rawSteer ? 0 : 1;
    return 0;
}
void cpuref_get_out(void* hh, pdb_step_out* o) { ((CpuRefHandle*)hh)->car.fillStepOut(*o); }
void cpuref_get_car_state(void* hh, pdb_car_state* cs) { ((CpuRefHandle*)hh)->car.fillCarState(*cs); }
float cpuref_env_gas(float a1) { return pdoracle::envGas(a1); }
// elementary functions of the portable-math specification (pm_ref.h) and of glibc, for tests/test_pmath.py
int cpuref_math_eval(int fn, int glibc, const float* x, const float* y, float* out, int n) {
    for (int i = 0; i < n; ++i) {
        const float a = x[i], b = y ? y[i] : 0.0f;
        if (glibc) {
            switch (fn) { case 0: out[i] = sinf(a); break; case 1: out[i] = cosf(a); break; case 2: out[i] = tanf(a); break; case 3: out[i] = atanf(a); break;
                          case 4: out[i] = atan2f(a, b); break; case 5: out[i] = asinf(a); break; case 6: out[i] = acosf(a); break; default: out[i] = powf(a, b); break; }
        } else {
            switch (fn) { case 0: out[i] = pmref::r_sinf(a); break; case 1: out[i] = pmref::r_cosf(a); break; case 2: out[i] = pmref::r_tanf(a); break; case 3: out[i] = pmref::r_atanf(a); break;
                          case 4: out[i] = pmref::r_atan2f(a, b); break; case 5: out[i] = pmref::r_asinf(a); break; case 6: out[i] = pmref::r_acosf(a); break; default: out[i] = pmref::r_powf(a, b); break; }
        }
    }
    return 0;
}
// Solver invariants (SURVEY.md 8c (ii),(iii)): the car's bodies and joints alone, no tyres / suspension forces, gravity off,
// launched as one rigid motion (chassis velocity v, spin w).  out: [0..2] linear momentum before, [3..5] after, [6..8] angular
// momentum about the origin before, [9..11] after, [12] kinetic energy before, [13] after, [14] worst distance-joint error (m),
// [15] worst ball / fixed anchor separation (m)
int cpuref_solver_freeflight(void* hh, int ticks, const float* v, const float* w, double* out) {
    auto* h = (CpuRefHandle*)hh;
    cpuref::Car car;
    car.init(&h->P, &h->T, h->s0);
    pdrb::World& W = car.w;
    W.gravity[0] = 0; W.gravity[1] = 0; W.gravity[2] = 0;
    const float* c = W.bodies[0].pos;
    for (auto& b : W.bodies) {
        const float r[3] = {b.pos[0] - c[0], b.pos[1] - c[1], b.pos[2] - c[2]};
        float wr[3]; pdrb::cross3(wr, w, r);
        for (int k = 0; k < 3; ++k) { b.lvel[k] = v[k] + wr[k]; b.avel[k] = w[k]; b.facc[k] = 0; b.tacc[k] = 0; }
    }
    auto measure = [&](double* P3, double* L3, double& E) {
        for (int k = 0; k < 3; ++k) { P3[k] = 0; L3[k] = 0; }
        E = 0;
        for (auto& b : W.bodies) {
            double Iw[9], t[9];
            for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) { t[r * 3 + cc] = 0; for (int k = 0; k < 3; ++k) t[r * 3 + cc] += (double)b.I[r * 3 + k] * b.R[cc * 3 + k]; }
            for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) { Iw[r * 3 + cc] = 0; for (int k = 0; k < 3; ++k) Iw[r * 3 + cc] += (double)b.R[r * 3 + k] * t[k * 3 + cc]; }
            double Lw[3];
            for (int r = 0; r < 3; ++r) Lw[r] = Iw[r * 3] * b.avel[0] + Iw[r * 3 + 1] * b.avel[1] + Iw[r * 3 + 2] * b.avel[2];
            const double m = b.mass, p[3] = {b.pos[0], b.pos[1], b.pos[2]}, vv[3] = {b.lvel[0], b.lvel[1], b.lvel[2]};
            for (int k = 0; k < 3; ++k) P3[k] += m * vv[k];
            L3[0] += m * (p[1] * vv[2] - p[2] * vv[1]) + Lw[0];
            L3[1] += m * (p[2] * vv[0] - p[0] * vv[2]) + Lw[1];
            L3[2] += m * (p[0] * vv[1] - p[1] * vv[0]) + Lw[2];
            E += 0.5 * m * (vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]) + 0.5 * (Lw[0] * b.avel[0] + Lw[1] * b.avel[1] + Lw[2] * b.avel[2]);
        }
    };
    measure(out + 0, out + 6, out[12]);
    double worstD = 0, worstA = 0;
    for (int t = 0; t < ticks; ++t) {
        W.step((float)(1.0 / 333.0));
        for (auto& j : W.joints) {
            float a1[3], a2[3];
            W.bodies[j.b0].relPointPos(j.anchor1, a1); W.bodies[j.b1].relPointPos(j.anchor2, a2);
            const double d = sqrt((double)(a1[0] - a2[0]) * (a1[0] - a2[0]) + (double)(a1[1] - a2[1]) * (a1[1] - a2[1]) + (double)(a1[2] - a2[2]) * (a1[2] - a2[2]));
            if (j.type == pdrb::JT_DBALL) worstD = std::max(worstD, fabs(d - (double)j.targetDistance));
            else if (j.type == pdrb::JT_BALL) worstA = std::max(worstA, d);
        }
    }
    measure(out + 3, out + 9, out[13]);
    out[14] = worstD; out[15] = worstA;
    return 0;
}
// the linear system of the last tick's rigid-body solve: A (m x m row-major, lower triangle meaningful), rhs and the fp32 solution
int cpuref_last_system(void* hh, float* A, float* rhs, float* lambda, int cap) {
    const pdrb::World& W = ((CpuRefHandle*)hh)->car.w;
    const int m = W.lastM;
    if (m > cap) return -1;
    if (A) memcpy(A, W.lastA.data(), sizeof(float) * m * m);
    if (rhs) memcpy(rhs, W.lastRhs.data(), sizeof(float) * m);
    if (lambda) memcpy(lambda, W.lastLambda.data(), sizeof(float) * m);
    return m;
}
// A lone box body (mass, box sides) with contact joints against the static world, stepped once by pdrb: for the closed-form
// checks of the contact rows in tests/test_contacts.py.  state = pos[3], lvel[3], avel[3]; contacts = n x pdb_contact;
// out = lvel[3], avel[3], pos[3] after the step, then lambda / lo / hi of the 3n contact rows
int cpuref_contact_unit(float mass, const float* sides, const float* state, const pdb_contact* contacts, int n, float dt, int gravityOn, float* out) {
    pdrb::World W;
    const int b = W.createBody();
    W.bodies[b].setMassBoxTotal(mass, sides[0], sides[1], sides[2]);
    for (int k = 0; k < 3; ++k) { W.bodies[b].pos[k] = state[k]; W.bodies[b].lvel[k] = state[3 + k]; W.bodies[b].avel[k] = state[6 + k]; }
    if (!gravityOn) { W.gravity[0] = 0; W.gravity[1] = 0; W.gravity[2] = 0; }
    W.contacts.resize(n);
    for (int i = 0; i < n; ++i) memcpy(&W.contacts[i], &contacts[i], sizeof(pdb_contact));
    W.contactBody = b;
    W.step(dt);
    for (int k = 0; k < 3; ++k) { out[k] = W.bodies[b].lvel[k]; out[3 + k] = W.bodies[b].avel[k]; out[6 + k] = W.bodies[b].pos[k]; }
    for (int i = 0; i < 3 * n; ++i) { out[9 + i] = W.lastContactLambda[i]; out[9 + 3 * n + i] = W.lastContactLo[i]; out[9 + 6 * n + i] = W.lastContactHi[i]; }
    return W.lastLcpIterations;
}
// Closed-form joint checks on two-body worlds (tests/test_physics_invariants.py): a heavy anchor body (1e9 kg, at rest at the
// origin, gravity off and applied by hand to the light body only) and a 1 kg bob.  kind 0: Ball joint at the origin, bob on a rod of
// length L released at angle theta0 -- returns the bob's x every tick (the test reads the swing period off it); kind 1: Slider along x,
// constant force F on the bob -- returns x, y, z and the relative rotation's vector part per tick; kind 2: DBall of length L holding
// the bob under gravity -- returns the anchor distance per tick.  out has ticks * 4 floats.
int cpuref_joint_unit(int kind, float L, float theta0, float F, int ticks, float* out) {
    pdrb::World W;
    W.gravity[0] = 0; W.gravity[1] = 0; W.gravity[2] = 0;
    const int a = W.createBody(), b = W.createBody();
    W.bodies[a].setMassBoxTotal(1.0e9f, 1, 1, 1);
    W.bodies[b].setMassBoxTotal(1.0f, 0.1f, 0.1f, 0.1f);
    const float g = 9.80665f, h = (float)(1.0 / 333.0);
    const float z3[3] = {0, 0, 0};
    if (kind == 0) {
        W.bodies[b].setPosition(L * sinf(theta0), -L * cosf(theta0), 0);
        const float c = cosf(theta0), s = sinf(theta0);
        const float R[9] = {c, -s, 0, s, c, 0, 0, 0, 1};   // rod along the body's -y
        W.bodies[b].setRotation(R);
        W.createBall(a, b, z3);
    } else if (kind == 1) {
        W.bodies[b].setPosition(0.5f, 0.2f, -0.1f);
        const float ax[3] = {1, 0, 0};
        W.createSlider(a, b, ax);
    } else {
        W.bodies[b].setPosition(0, -L, 0);
        const float p2[3] = {0, -L, 0};
        W.createDBall(a, b, z3, p2);
    }
    for (int t = 0; t < ticks; ++t) {
        pdrb::Body& B = W.bodies[b];
        if (kind == 1) { const float f[3] = {F, 0, 0}; B.addForceAtPos(f, B.pos); }
        else { const float f[3] = {0, -g * B.mass, 0}; B.addForceAtPos(f, B.pos); }
        W.step(h);
        float* o = out + 4 * t;
        if (kind == 0) { o[0] = B.pos[0]; o[1] = B.pos[1]; o[2] = sqrtf(B.pos[0] * B.pos[0] + B.pos[1] * B.pos[1] + B.pos[2] * B.pos[2]); o[3] = 0; }
        else if (kind == 1) { o[0] = B.pos[0]; o[1] = B.pos[1]; o[2] = B.pos[2]; o[3] = sqrtf(B.q[1] * B.q[1] + B.q[2] * B.q[2] + B.q[3] * B.q[3]); }
        else { o[0] = sqrtf(B.pos[0] * B.pos[0] + B.pos[1] * B.pos[1] + B.pos[2] * B.pos[2]); o[1] = B.pos[0]; o[2] = B.pos[2]; o[3] = W.bodies[a].pos[1]; }
    }
    return 0;
}
const char* cpuref_scenario_name(int sid) { return pdoracle::kScenarios[sid].name; }
const char* cpuref_scenario_track(int sid) { return pdoracle::kScenarios[sid].track; }
const char* cpuref_scenario_car(int sid) { return pdoracle::kScenarios[sid].car ? pdoracle::kScenarios[sid].car : PDORACLE_DEFAULT_CAR; }
int cpuref_num_scenarios(void) { return pdoracle::kNumScenarios; }
int cpuref_scenario_collide(int sid) { return pdoracle::kScenarios[sid].collide; }
int cpuref_scenario_auto_teleport(int sid) { return pdoracle::kScenarios[sid].autoTele; }
int cpuref_scenario_resets(int sid) { return pdoracle::kScenarios[sid].resetEvery; }
// the rest of a scenario's script, for drivers that step something else through it (tests/scenario_util.py drives the GPU):
// out = {resetEvery, teleDist, boostAt, feedback, collide, autoTele, stride, denseTicks}
void cpuref_scenario_fields(int sid, int* out) {
    const auto& sc = pdoracle::kScenarios[sid];
    out[0] = sc.resetEvery; out[1] = sc.teleDist; out[2] = sc.boostAt; out[3] = sc.feedback; out[4] = sc.collide; out[5] = sc.autoTele; out[6] = sc.stride; out[7] = sc.denseTicks;
}
float cpuref_scenario_teledist(int k) { return pdoracle::kTeleDist[k & 3]; }
// the k-th mid-run teleport of a scenario as (kind, a, b, c): kind 0 = teleportCarToSpline(a) (a = 0: the start), 1 = teleportCarToPits((int)a),
// 2 = teleportCarToLocation(chassis position + (a, b, c))
int cpuref_scenario_teleport(int sid, int k, float* abc) {
    const auto& sc = pdoracle::kScenarios[sid];
    abc[0] = abc[1] = abc[2] = 0.0f;
    if (sc.teleDist == 1) { abc[0] = pdoracle::kTeleDist[k % 4]; return 0; }
    if (sc.teleDist == 2) { abc[0] = (float)pdoracle::kTelePit[k % 5]; return 1; }
    if (sc.teleDist == 3) { for (int i = 0; i < 3; ++i) abc[i] = pdoracle::kTeleLoc[k % 4][i]; return 2; }
    return 0;
}
void cpuref_scenario_action(int sid, int tick, float* a) { pdoracle::scenarioAction(sid, tick, a[0], a[1]); }
int cpuref_scenario_scoring(int sid, int i, const char** name, float* value) {
    if (!pdoracle::kScenarios[sid].scoringSet || i < 0 || i >= pdoracle::kNumScoringSetA) return 0;
    *name = pdoracle::kScoringSetA[i].name; *value = pdoracle::kScoringSetA[i].value;
    return 1;
}
// i-th setCarTune call of the scenario's tune set: returns 0 past the end
int cpuref_scenario_tune(int sid, int i, const char** name, float* value) {
    if (!pdoracle::kScenarios[sid].tuneSet || i < 0 || i >= pdoracle::kNumTuneSetA) return 0;
    *name = pdoracle::kTuneSetA[i].name; *value = pdoracle::kTuneSetA[i].value;
    return 1;
}
void cpuref_scenario_feedback(int sid, int tick, const float* obs, float* a) { pdoracle::scenarioFeedback(sid, tick, obs, a[0], a[1]); }

// run one scripted scenario exactly like oracle/refharness/ref_main.cpp and write the probe file
int cpuref_run_scenario_cb(void* hh, int sid, const char* outPath, void (*teleport)(pdb_dyn_state*, int, float, float, float));
int cpuref_run_scenario(void* hh, int sid, const char* outPath) { return cpuref_run_scenario_cb(hh, sid, outPath, nullptr); }
// teleport(state, kind, a, b, c): the mid-run teleports on a state record -- the PRODUCT's host functions, handed in by the test (kind 0:
// pdb_teleport_to_spline(a) = Car::teleportByMode(Start) for a = 0; 1: pdb_teleport_to_pit((int)a); 2: pdb_teleport_to_location(a, b, c)), so
// that the scenarios with mid-run resets pin them against the reference's own Car::teleportToSpline / teleportToPits / forcePosition
int cpuref_run_scenario_cb(void* hh, int sid, const char* outPath, void (*teleport)(pdb_dyn_state*, int, float, float, float)) {
    auto* h = (CpuRefHandle*)hh;
    const auto& sc = pdoracle::kScenarios[sid];
    // setCarAssists (PyProjectD.cpp:307-317) per scenario; smooth steering stays on like the env
    h->P.acUseOnStart = sc.autoClutch; h->P.acUseOnChange = sc.autoClutch; h->P.autoShiftActive = sc.autoShift; h->P.autoBlipActive = sc.autoBlip;
    h->P.smoothSteer = sc.rawSteer ? 0 : 1;
    h->car = cpuref::Car();
    h->car.init(&h->P, &h->T, h->s0);
    h->car.autoTeleportHook = h->hook;
    pdoracle::ProbeFile pf;
    h->car.step(0.0f, pdoracle::envGas(0.0f), (float)(1.0 / 333.0), 1.0 / 333.0);   // env.reset(): teleport (already in s0) + step([0,0])
    { pdoracle::Probe P; P.names = &pf.names; h->car.fillProbe(P); pf.add(-1, 0.0f, 0.0f, P); }
    for (int t = 0; t < sc.ticks; ++t) {
        if (sc.resetEvery && t > 0 && t % sc.resetEvery == 0) {   // env.reset(): teleportCarByMode(Start) + step([0,0])
            if (!teleport) return -2;
            pdb_dyn_state st = h->car.S;
            float abc[3];
            const int kind = cpuref_scenario_teleport(sid, t / sc.resetEvery - 1, abc);
            if (kind == 2) for (int i = 0; i < 3; ++i) abc[i] = st.body[PDB_BODY_CHASSIS].pos[i] + abc[i];
            teleport(&st, kind, abc[0], abc[1], abc[2]);
            h->car.loadState(st);
            h->car.step(0.0f, pdoracle::envGas(0.0f), (float)(1.0 / 333.0), 1.0 / 333.0);
        }
        if (sc.boostAt && t == sc.boostAt) {
            pdb_dyn_state st = h->car.S;
            for (int b = 0; b < h->P.numBodies; ++b) st.body[b].lvel[2] = 50.0f;
            h->car.loadState(st);
        }
        float a0, a1;
        if (sc.feedback) {
            pdb_step_out o; h->car.fillStepOut(o);
            pdoracle::scenarioFeedback(sid, t, o.obs, a0, a1);
            h->car.step(a0, pdoracle::envGas(a1), (float)(1.0 / 333.0), 1.0 / 333.0);
        } else if (sc.full) {
            float a[8]; cpuref_scenario_controls(sid, t, a);
            a0 = a[0]; a1 = a[4];
            h->car.stepControls(ctlOf(a), (float)(1.0 / 333.0), 1.0 / 333.0);
        } else {
            pdoracle::scenarioAction(sid, t, a0, a1);
            h->car.step(a0, pdoracle::envGas(a1), (float)(1.0 / 333.0), 1.0 / 333.0);
        }
        if (pdoracle::scenarioRecord(sc, t)) { pdoracle::Probe P; h->car.fillProbe(P); pf.add(t, a0, a1, P); }
    }
    return pf.write(outPath) ? 0 : -1;
}

// A two-car scenario (oracle/scenarios.h twoCar) exactly like oracle/refharness/ref_main.cpp: two cars of one simulator -- h0 the first (at the start), h1 the second
// (created from a record already put down kTwoCarDist ahead) -- stepped tick by tick, each reading the slipstream the OTHER's last tick left (every car of a Simulator
// steps before any postStep: Simulator.cpp:168-201); one probe file per car
int cpuref_run_scenario2(void* hh0, void* hh1, int sid, const char* outPath0, const char* outPath1) {
    CpuRefHandle* h[2] = {(CpuRefHandle*)hh0, (CpuRefHandle*)hh1};
    const auto& sc = pdoracle::kScenarios[sid];
    if (!sc.twoCar || sc.resetEvery || sc.boostAt || sc.full) return -3;
    for (int c = 0; c < 2; ++c) {
        h[c]->P.acUseOnStart = sc.autoClutch; h[c]->P.acUseOnChange = sc.autoClutch; h[c]->P.autoShiftActive = sc.autoShift; h[c]->P.autoBlipActive = sc.autoBlip;
        h[c]->P.smoothSteer = sc.rawSteer ? 0 : 1;
        h[c]->car = cpuref::Car();
        h[c]->car.init(&h[c]->P, &h[c]->T, h[c]->s0);
        h[c]->car.physicsGUID = c;
    }
    pdoracle::ProbeFile pf[2];
    const float dt = (float)(1.0 / 333.0); const double dtD = 1.0 / 333.0;
    auto stepBoth = [&](float a0, float a1, float b0, float b1) {
        const pdb_slip_state s0 = h[0]->car.slip, s1 = h[1]->car.slip;
        h[0]->car.otherSlips.assign(1, s1); h[1]->car.otherSlips.assign(1, s0);
        h[0]->car.step(a0, pdoracle::envGas(a1), dt, dtD);
        h[1]->car.step(b0, pdoracle::envGas(b1), dt, dtD);
    };
    stepBoth(0.0f, 0.0f, 0.0f, 0.0f);   // env.reset(): the teleports (already in the records) + step([0,0])
    for (int c = 0; c < 2; ++c) { pdoracle::Probe P; P.names = &pf[c].names; h[c]->car.fillProbe(P); pf[c].add(-1, 0.0f, 0.0f, P); }
    for (int t = 0; t < sc.ticks; ++t) {
        float a0, a1, b0, b1;
        if (sc.feedback) {
            pdb_step_out o; h[0]->car.fillStepOut(o); pdoracle::scenarioFeedback(sid, t, o.obs, a0, a1);
            h[1]->car.fillStepOut(o); pdoracle::scenarioFeedback2(sid, t, o.obs, b0, b1);
        } else { pdoracle::scenarioAction(sid, t, a0, a1); pdoracle::scenarioAction2(sid, t, b0, b1); }
        stepBoth(a0, a1, b0, b1);
        if (pdoracle::scenarioRecord(sc, t)) {
            pdoracle::Probe P; h[0]->car.fillProbe(P); pf[0].add(t, a0, a1, P);
            pdoracle::Probe Q; h[1]->car.fillProbe(Q); pf[1].add(t, b0, b1, Q);
        }
    }
    return (pf[0].write(outPath0) && pf[1].write(outPath1)) ? 0 : -1;
}
int cpuref_scenario_two_car(int sid, float* dist) { const int k = pdoracle::kScenarios[sid].twoCar; if (k && dist) *dist = pdoracle::kTwoCarDist[k - 1]; return k; }
void cpuref_scenario_action2(int sid, int tick, float* a) { pdoracle::scenarioAction2(sid, tick, a[0], a[1]); }
void cpuref_scenario_feedback2(int sid, int tick, const float* obs, float* a) { pdoracle::scenarioFeedback2(sid, tick, obs, a[0], a[1]); }

// CPU baseline: step `ncars` independent cars for `ticks` ticks with per-car constant env actions;
// returns elapsed seconds.  threads <= 1: single thread.
double cpuref_bench(void* hh, int ncars, int ticks, const float* actions, int threads, pdb_step_out* lastOut) {
    auto* h = (CpuRefHandle*)hh;
    std::vector<cpuref::Car> cars(ncars);
    for (int i = 0; i < ncars; ++i) cars[i].init(&h->P, &h->T, h->s0);
    const auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
    if (threads > 1) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) if (threads > 1)
#endif
    for (int i = 0; i < ncars; ++i) {
        const float steer = actions[2 * i], gas = pdoracle::envGas(actions[2 * i + 1]);
        for (int t = 0; t < ticks; ++t) cars[i].step(steer, gas, (float)(1.0 / 333.0), 1.0 / 333.0);
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (lastOut) for (int i = 0; i < ncars; ++i) cars[i].fillStepOut(lastOut[i]);
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
