// pdbatch C ABI, host-side entry points (content loading, tunes, reset poses).  See include/pdbatch.h.
#include "pdbatch.h"
#include "model.hpp"
#include "../device/pmath.hpp"
#include <cstdlib>
#include <cstring>
#include <string>

namespace pdb {
thread_local std::string g_lastError;
void setError(const std::string& s) { g_lastError = s; }
}

#define PDB_TRY try {
#define PDB_CATCH(code) } catch (const std::exception& e) { pdb::setError(e.what()); return code; } catch (...) { pdb::setError("unknown error"); return code; }

extern "C" {

const char* pdb_last_error(void) { return pdb::g_lastError.c_str(); }
const char* pdb_version(void) { return "pdbatch 0.1 (gfx950)"; }

int pdb_build_car_model(const char* base_path, const char* model_name, pdb_car_params* out) {
    if (!base_path || !model_name || !out) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::buildCarModel(base_path, model_name, *out);
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_set_car_tune(pdb_car_params* params, const char* base_path, const char* model_name, const char* name, float value, int raw) {
    if (!params || !base_path || !model_name || !name) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::setCarTune(*params, base_path, model_name, name, value, raw != 0);   // unknown names are ignored like the reference
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_set_scoring_var(pdb_car_params* params, const char* name, float value) {
    if (!params || !name) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (!pdb::setScoringVar(*params, name, value)) { pdb::setError(std::string("unknown scoring var ") + name); return PDB_ERR_ARG; }
    return PDB_OK;
}
int pdb_lane_tune_from_params(const pdb_car_params* params, pdb_lane_tune* row) {
    if (!params || !row) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    memset(row, 0, sizeof(*row));
    row->finalRatio = params->finalRatio; row->diffPowerRamp = params->diffPowerRamp; row->diffCoastRamp = params->diffCoastRamp;
    row->frontBias = params->frontBias;
    for (int i = 0; i < 4; ++i) row->pressureStatic[i] = params->tyre[i].pressureStatic;
    row->scoring = params->scoring;
    row->valid = 1;
    return PDB_OK;
}
int pdb_lane_setup_from_params(const pdb_car_params* params, pdb_lane_setup* row) {
    if (!params || !row) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    memset(row, 0, sizeof(*row));
    row->diffPreLoad = params->diffPreLoad;
    for (int g = 0; g < PDB_MAX_GEARS; ++g) row->gearRatio[g] = params->gearRatio[g];
    row->brakePowerMultiplier = params->brakePowerMultiplier; row->limiterMultiplier = params->limiterMultiplier;
    row->arbK[0] = params->arbK[0]; row->arbK[1] = params->arbK[1];
    for (int t = 0; t < PDB_MAX_TURBOS; ++t) row->turboUserSetting[t] = params->turbos[t].userSetting;
    for (int i = 0; i < 4; ++i) {
        const pdb_susp& su = params->susp[i];
        pdb_lane_wheel& w = row->wheel[i];
        w.bumpFast = su.damper.bumpFast; w.bumpSlow = su.damper.bumpSlow; w.reboundFast = su.damper.reboundFast; w.reboundSlow = su.damper.reboundSlow;
        w.bumpStopRate = su.bumpStopRate; w.k = su.k; w.progressiveK = su.progressiveK; w.rodLength = su.rodLength; w.packerRange = su.packerRange; w.toeOutLinear = su.toeOutLinear;
        // the camber rotation as the tick uses it (mat44f::createFromAxisAngle((0,0,1), staticCamber): the expressions of the batch's constants block, csrc/device/batch.hip fillConst)
        const float s = pm::sinf_(su.staticCamber), c = pm::cosf_(su.staticCamber), o = 1.0f - c;
        w.camC = ((0.0f * 0.0f) * o) + c;
        w.camS = (1.0f * s) + (0.0f * 0.0f) * o;
        w.camM33 = ((1.0f * 1.0f) * o) + c;
    }
    return PDB_OK;
}
float pdb_get_scoring_var(const pdb_car_params* params, const char* name) {
    float w = 0;
    if (params && name) pdb::getScoringVar(*params, name, w);
    return w;
}
int pdb_set_assists(pdb_car_params* params, int auto_clutch, int auto_shift, int auto_blip, int smooth_steer) {
    if (!params) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    params->acUseOnStart = auto_clutch ? 1 : 0;
    params->acUseOnChange = auto_clutch ? 1 : 0;
    params->autoShiftActive = auto_shift ? 1 : 0;
    params->autoBlipActive = auto_blip ? 1 : 0;
    params->smoothSteer = smooth_steer ? 1 : 0;
    return PDB_OK;
}
int pdb_build_track(const char* base_path, const char* track_name, void** blob, uint64_t* bytes) {
    return pdb_build_track_opts(base_path, track_name, 0, blob, bytes);
}
int pdb_build_track_opts(const char* base_path, const char* track_name, int flags, void** blob, uint64_t* bytes) {
    if (!base_path || !track_name || !blob || !bytes) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (flags & ~PDB_TRACK_RECOMPUTE_FAT_POINTS) { pdb::setError("unknown track build flag"); return PDB_ERR_ARG; }
    PDB_TRY
    std::vector<uint8_t> v = pdb::buildTrack(base_path, track_name, (flags & PDB_TRACK_RECOMPUTE_FAT_POINTS) != 0);
    void* p = malloc(v.size());
    if (!p) { pdb::setError("out of memory"); return PDB_ERR_ARG; }
    memcpy(p, v.data(), v.size());
    *blob = p; *bytes = v.size();
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
void pdb_free(void* p) { free(p); }
int pdb_initial_state(const pdb_car_params* params, const void* track_blob, pdb_dyn_state* out) {
    if (!params || !track_blob || !out) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::TrackView tv(static_cast<const uint8_t*>(track_blob));
    pdb::initialState(*params, tv, *out);
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_teleport_to_spline(const pdb_car_params* params, const void* track_blob, float distance_norm, pdb_dyn_state* inout) {
    if (!params || !track_blob || !inout) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::TrackView tv(static_cast<const uint8_t*>(track_blob));
    pdb::teleportToSpline(*params, tv, distance_norm, *inout);
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_teleport_to_pit(const pdb_car_params* params, const void* track_blob, int pit_id, pdb_dyn_state* inout) {
    if (!params || !track_blob || !inout) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::TrackView tv(static_cast<const uint8_t*>(track_blob));
    pdb::teleportToPit(*params, tv, pit_id, *inout);   // an id outside the list: nothing happens, like Car::teleportToPits
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_teleport_to_location(const pdb_car_params* params, const void* track_blob, float x, float y, float z, pdb_dyn_state* inout) {
    if (!params || !track_blob || !inout) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::TrackView tv(static_cast<const uint8_t*>(track_blob));
    const float pos[3] = {x, y, z};
    pdb::teleportToLocation(*params, tv, pos, *inout);
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_track_num_pits(const void* track_blob) {
    if (!track_blob) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    const pdb_track_header* h = static_cast<const pdb_track_header*>(track_blob);
    if (h->magic != 0x4B544450 || h->version != 6) { pdb::setError("pdb_track_num_pits: not a track blob of this version"); return PDB_ERR_ARG; }
    return h->numPits;
}
int pdb_track_pit(const void* track_blob, int pit_id, float* m16) {
    const int n = pdb_track_num_pits(track_blob);
    if (n < 0) return n;
    if (!m16 || pit_id < 0 || pit_id >= n) { pdb::setError("pdb_track_pit: bad argument"); return PDB_ERR_ARG; }
    const pdb_track_header* h = static_cast<const pdb_track_header*>(track_blob);
    memcpy(m16, static_cast<const uint8_t*>(track_blob) + h->offPits + (size_t)pit_id * 64, 64);
    return PDB_OK;
}

int pdb_teleport_by_mode(const pdb_car_params* params, const void* track_blob, int mode, pdb_dyn_state* inout) {
    if (!params || !track_blob || !inout || mode < 0 || mode > 2) { pdb::setError("pdb_teleport_by_mode: bad argument"); return PDB_ERR_ARG; }
    PDB_TRY
    pdb::TrackView tv(static_cast<const uint8_t*>(track_blob));
    pdb::teleportByMode(*params, tv, mode, *inout);
    return PDB_OK;
    PDB_CATCH(PDB_ERR_IO)
}
int pdb_set_auto_teleport(pdb_car_params* params, int on_collision, int on_bad_location, int mode) {
    if (!params || mode < 0 || mode > 2) { pdb::setError("pdb_set_auto_teleport: bad argument"); return PDB_ERR_ARG; }
    params->autoTeleport = (on_collision ? 1 : 0) | (on_bad_location ? 2 : 0) | (mode << 2);
    return PDB_OK;
}

// reproducible elementary functions of the step (device/pmath.hpp compiled for the host): lets a maintainer check,
// on any platform, that host and device evaluate them identically (the kernel's constants come from these too)
int pdb_math_eval(int fn, const float* x, const float* y, float* out, int n) {
    if (!x || !out || n < 0 || fn < 0 || fn > 7 || ((fn == 4 || fn == 7) && !y)) { pdb::setError("pdb_math_eval: bad argument"); return PDB_ERR_ARG; }
    for (int i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = pm::sinf_(x[i]); break;
            case 1: out[i] = pm::cosf_(x[i]); break;
            case 2: out[i] = pm::tanf_(x[i]); break;
            case 3: out[i] = pm::atanf_(x[i]); break;
            case 4: out[i] = pm::atan2f_(x[i], y[i]); break;
            case 5: out[i] = pm::asinf_(x[i]); break;
            case 6: out[i] = pm::acosf_(x[i]); break;
            default: out[i] = pm::powf_(x[i], y[i]); break;
        }
    }
    return PDB_OK;
}

}  // extern "C"
