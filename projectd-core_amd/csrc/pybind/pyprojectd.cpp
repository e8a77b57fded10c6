// PyProjectD-compatible python module over the pdbatch C ABI (include/pdbatch.h).
//
// Same module name, classes, functions, argument meaning and error convention as the reference's pybind11 module
// (reference src/PyProjectD/PyProjectD.cpp:515-640), so pyprojectd/projectd_env.py runs against it unchanged:
//   * creators return -1 on failure and log; everything else silently ignores unknown ids (:74-109)
//   * nothing throws into python; the GIL is never released
//   * a simulator holds one track and its cars (one in every env: projectd_env.py:118-121; up to cfg/sim.ini's MAX_CARS of one model, coupled through the
//     slipstream -- no body contacts between cars); they are the lanes of one world of a device batch.  The classic per-simulator calls drive a 1-car batch; createBatch() widens a configured simulator
//     into N identical lanes stepped by one kernel launch (stepBatch), which is the point of this build.
// The playground / window functions (:600-640) exist and do nothing: rendering is outside the hot path.
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "pdbatch.h"

namespace py = pybind11;

namespace {

#pragma pack(push, 4)
struct vec3f { float x = 0, y = 0, z = 0; };
struct mat44f { float M11 = 1, M12 = 0, M13 = 0, M14 = 0, M21 = 0, M22 = 1, M23 = 0, M24 = 0, M31 = 0, M32 = 0, M33 = 1, M34 = 0, M41 = 0, M42 = 0, M43 = 0, M44 = 1; };
struct CarControls {   // reference Car/CarControls.h
    float steer = 0, clutch = 0, brake = 0, handBrake = 0, gas = 0;
    int8_t isShifterSupported = 1, requestedGearIndex = -1, gearUp = 0, gearDn = 0;
};
struct CarState {      // reference Car/CarState.h (664 bytes, pack 4) == pdb_car_state
    int32_t carId = 0, simId = 0; float timestamp = 0; CarControls controls;
    int32_t collisionFlag = 0, outOfTrackFlag = 0, trackPointId = 0; float lastTrackPointTimestamp = 0, trackLocation = 0, bodyVsTrack = 0, velocityVsTrack = 0;
    float engineRPM = 0, speedMS = 0; int32_t gear = 0, gearGrinding = 0;
    mat44f bodyMatrix; vec3f bodyPos, bodyEuler, accG, velocity, localVelocity, angularVelocity, localAngularVelocity;
    std::array<mat44f, 4> hubMatrix; std::array<vec3f, 4> tyreContacts;
    std::array<float, 4> tyreLoad{}, tyreAngularSpeed{}, tyreSlipRatio{}, tyreNdSlip{};
    std::array<float, 10> probes{}; std::array<float, 5> lookAhead{};
    float stepReward = 0, totalReward = 0;
};
#pragma pack(pop)
static_assert(sizeof(CarState) == sizeof(pdb_car_state), "CarState must match the reference layout");
static_assert(sizeof(CarControls) == sizeof(pdb_controls), "CarControls must match the reference layout");

std::mutex g_lock;             // protects the maps only, like SIM_LOCK (:16-17)
std::string g_logFile;

void logf(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (!g_logFile.empty()) { if (FILE* f = fopen(g_logFile.c_str(), "a")) { fprintf(f, "%s\n", buf); fclose(f); } }
    else if (getenv("PDB_VERBOSE")) fprintf(stderr, "%s\n", buf);
}

// A car of a simulator beyond the first (Simulator::addCar again: cfg/sim.ini MAX_CARS, 1..100; PyProjectD.cpp:219-237).  The cars of a simulator are the lanes of ONE
// world of its device batch (pdb_set_world_size): same model, each with its own block where its tunes / scoring variables / assists differ (per-lane rows), coupled through
// the slipstream.  Body contacts between cars are not built.
struct ExtraCar {
    pdb_car_params P{};
    pdb_dyn_state S{};
    pdb_contact contacts[PDB_MAX_CONTACTS] = {};
    CarControls controls;
    CarState state;
};
struct Sim {
    int id = 0;
    std::string base, trackName, model;
    std::vector<uint8_t> track;
    bool hasCar = false;
    pdb_car_params P{};           // car 0
    pdb_dyn_state S{};            // host copy; authoritative while batch == nullptr
    std::vector<std::unique_ptr<ExtraCar>> more;   // cars 1 ..
    std::vector<pdb_slip_state> slips;             // [2][cars]: the cars' wakes (state of a multi-car simulator), host copy
    int maxCars = 1;              // cfg/sim.ini [SIM] MAX_CARS, clamped to 1..100 (Simulator.cpp:59-60)
    pdb_batch* batch = nullptr;   // device batch of the simulator's cars (one world), created at the first step
    bool paramsDirty = false;
    CarControls controls;
    CarState state;
    double physicsTime = 0;
    int device = 0;
    ~Sim() { if (batch) pdb_destroy(batch); }

    pdb_contact contacts[PDB_MAX_CONTACTS] = {};   // car 0's live contact joints (the first S.numContacts): they move with the record
    int numCars() const { return hasCar ? 1 + (int)more.size() : 0; }
    pdb_car_params& carP(int c) { return c == 0 ? P : more[(size_t)c - 1]->P; }
    pdb_dyn_state& carS(int c) { return c == 0 ? S : more[(size_t)c - 1]->S; }
    pdb_contact* carContacts(int c) { return c == 0 ? contacts : more[(size_t)c - 1]->contacts; }
    CarControls& carControls(int c) { return c == 0 ? controls : more[(size_t)c - 1]->controls; }
    CarState& carState(int c) { return c == 0 ? state : more[(size_t)c - 1]->state; }
    bool pullState() {
        if (!batch) return true;
        const int n = numCars();
        for (int c = 0; c < n; ++c)
            if (pdb_get_state(batch, c, 1, &carS(c)) != PDB_OK || pdb_get_contacts(batch, c, 1, carContacts(c)) != PDB_OK) return false;
        if (n > 1) { slips.resize(2 * (size_t)n); if (pdb_get_slipstreams(batch, 0, n, slips.data()) != PDB_OK) return false; }
        return true;
    }
    bool pushState() {
        if (!batch) return true;
        const int n = numCars();
        for (int c = 0; c < n; ++c)
            if (pdb_set_state(batch, c, 1, &carS(c)) != PDB_OK || pdb_set_contacts(batch, c, 1, carContacts(c)) != PDB_OK) return false;
        if (n > 1 && slips.size() == 2 * (size_t)n && pdb_set_slipstreams(batch, 0, n, slips.data()) != PDB_OK) return false;
        return true;
    }
    bool ensureBatch() {
        if (batch && paramsDirty) { pullState(); pdb_destroy(batch); batch = nullptr; }
        if (!batch) {
            const int n = numCars();
            batch = pdb_create(device, n, &P, track.data(), track.size(), PDB_ACTION_FULL);
            if (!batch) { logf("EXCEPTION: %s", pdb_last_error()); return false; }
            if (n > 1) {
                if (pdb_set_world_size(batch, n) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return false; }
                for (int c = 1; c < n; ++c) {   // a car whose block differs from car 0's (its own setCarTune / setScoringVar calls): its lane's rows
                    if (memcmp(&carP(c), &P, sizeof(P)) == 0) continue;
                    pdb_lane_tune lt; pdb_lane_setup ls;
                    if (pdb_lane_tune_from_params(&carP(c), &lt) != PDB_OK || pdb_set_lane_tunes(batch, c, 1, &lt) != PDB_OK ||
                        pdb_lane_setup_from_params(&carP(c), &ls) != PDB_OK || pdb_set_lane_setups(batch, c, 1, &ls) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return false; }
                }
            }
            if (!pushState()) { logf("EXCEPTION: %s", pdb_last_error()); return false; }
        }
        paramsDirty = false;
        return true;
    }
    template <class F> void editState(int c, F f) {   // a host-side edit of one car's record (teleports): pull, edit, push
        if (!hasCar || track.empty() || c < 0 || c >= numCars()) return;
        pullState();
        if (f(&carP(c), &carS(c)) != PDB_OK) logf("EXCEPTION: %s", pdb_last_error());
        pushState();
    }
    void teleportSpline(int c, float d) { editState(c, [&](pdb_car_params* p, pdb_dyn_state* st) { return pdb_teleport_to_spline(p, track.data(), d, st); }); }
    // Car::teleportToPits (Car.cpp:1316-1323): an id outside pits.ini's list leaves the car alone
    void teleportPit(int c, int pitId) { editState(c, [&](pdb_car_params* p, pdb_dyn_state* st) { return pdb_teleport_to_pit(p, track.data(), pitId, st); }); }
    // Car::forcePosition (Car.cpp:1240-1272)
    void teleportLocation(int c, float x, float y, float z) { editState(c, [&](pdb_car_params* p, pdb_dyn_state* st) { return pdb_teleport_to_location(p, track.data(), x, y, z, st); }); }
    // Car::teleportByMode (Car.cpp:1320-1336); Random draws from the car's own C-runtime rand() state
    void teleportByMode(int c, int mode) { if (mode < 0 || mode > 2) return; editState(c, [&](pdb_car_params* p, pdb_dyn_state* st) { return pdb_teleport_by_mode(p, track.data(), mode, st); }); }
};

struct Batch {
    int id = 0, n = 0;
    pdb_batch* b = nullptr;
    std::vector<pdb_step_out> out;
    ~Batch() { if (b) pdb_destroy(b); }
};

std::unordered_map<int, std::shared_ptr<Sim>> g_sims;
std::unordered_map<int, std::shared_ptr<Batch>> g_batches;
int g_uniqSimId = 0, g_uniqBatchId = 0;

Sim* getSim(int simId) { auto it = g_sims.find(simId); return it == g_sims.end() ? nullptr : it->second.get(); }
Sim* getCarSim(int simId, int carId) { Sim* s = getSim(simId); return (s && s->hasCar && carId >= 0 && carId < s->numCars()) ? s : nullptr; }
Batch* getBatch(int id) { auto it = g_batches.find(id); return it == g_batches.end() ? nullptr : it->second.get(); }

// ---- logging / seed (:50-68) ----
// srand of the process' C runtime in the reference; here every simulator (and every lane of a batch made from it) carries its
// own generator state in its record: setSeed seeds the simulators that exist and the ones created afterwards
unsigned int g_seed = 1;
void setSeed(unsigned int seed);
void setLogFile(const std::string& filename, bool overwrite) { g_logFile = filename; if (overwrite) { if (FILE* f = fopen(filename.c_str(), "w")) fclose(f); } }
void clearLogFile() { if (!g_logFile.empty()) { if (FILE* f = fopen(g_logFile.c_str(), "w")) fclose(f); } }
void writeLog(const std::string& msg) { logf("%s", msg.c_str()); }

void setSeed(unsigned int seed) {
    g_seed = seed;
    for (auto& kv : g_sims) { Sim* s = kv.second.get(); if (s->hasCar) { s->pullState(); for (int c = 0; c < s->numCars(); ++c) s->carS(c).randState = (int32_t)seed; s->pushState(); } }
}
// ---- simulator (:111-180) ----
int createSimulator(const std::string& basePath) {
    std::ifstream ini(basePath + "/cfg/sim.ini");
    if (!ini.good()) { logf("EXCEPTION: cannot open %s/cfg/sim.ini", basePath.c_str()); return -1; }
    auto s = std::make_shared<Sim>();
    s->base = basePath;
    {   // [SIM] MAX_CARS (Simulator.cpp:59-60: clamped to 1..100; the shipped cfg/sim.ini says 2)
        std::string line; bool inSim = false;
        while (std::getline(ini, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (!line.empty() && line[0] == '[') inSim = line.rfind("[SIM]", 0) == 0;
            else if (inSim && line.rfind("MAX_CARS=", 0) == 0) { const int v = atoi(line.c_str() + 9); s->maxCars = v < 1 ? 1 : (v > 100 ? 100 : v); }
        }
    }
    if (const char* d = getenv("PDB_DEVICE")) s->device = atoi(d);
    std::lock_guard<std::mutex> g(g_lock);
    s->id = g_uniqSimId++;
    logf("[PY] createSimulator simId=%d", s->id);
    g_sims.insert({s->id, s});
    return s->id;
}
void destroySimulator(int simId) { logf("[PY] destroySimulator simId=%d", simId); std::lock_guard<std::mutex> g(g_lock); g_sims.erase(simId); }
void destroyAllSimulators() { std::lock_guard<std::mutex> g(g_lock); g_sims.clear(); g_batches.clear(); g_uniqSimId = 0; }

void stepSimulator(int simId, double dt) {
    Sim* s = getSim(simId);
    if (!s || !s->hasCar) return;
    if (!s->ensureBatch()) return;
    const int n = s->numCars();
    std::vector<float> a(8 * (size_t)n);
    for (int k = 0; k < n; ++k) {
        const CarControls& c = s->carControls(k);
        const float row[8] = {c.steer, c.clutch, c.brake, c.handBrake, c.gas, (float)c.requestedGearIndex, (float)(c.gearUp != 0), (float)(c.gearDn != 0)};
        memcpy(&a[8 * (size_t)k], row, sizeof(row));
    }
    std::vector<pdb_step_out> o((size_t)n);
    if (pdb_step_host(s->batch, a.data(), (float)dt, o.data()) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return; }
    for (int k = 0; k < n; ++k) {
        pdb_car_state cs;
        if (pdb_get_car_state(s->batch, k, 1, &cs) == PDB_OK) { CarState& st = s->carState(k); memcpy(&st, &cs, sizeof(cs)); st.carId = k; st.simId = s->id; }
    }
    s->physicsTime += dt;   // (setCarAutoTeleport: the teleport happens inside the tick, in the kernel, like ScoringSystem.cpp:194-226)
}

// ---- track (:186-215) ----
void loadTrack(int simId, const std::string& trackName) {
    logf("[PY] loadTrack simId=%d trackName=%s", simId, trackName.c_str());
    Sim* s = getSim(simId);
    if (!s) return;
    void* blob = nullptr; uint64_t bytes = 0;
    if (pdb_build_track(s->base.c_str(), trackName.c_str(), &blob, &bytes) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return; }
    s->track.assign((uint8_t*)blob, (uint8_t*)blob + bytes);
    pdb_free(blob);
    s->trackName = trackName;
    if (s->batch) { pdb_destroy(s->batch); s->batch = nullptr; }
    for (int c = 0; c < s->numCars(); ++c) if (pdb_initial_state(&s->carP(c), s->track.data(), &s->carS(c)) != PDB_OK) logf("EXCEPTION: %s", pdb_last_error());
    s->slips.clear();
}
void unloadTrack(int simId) { Sim* s = getSim(simId); if (!s) return; if (s->batch) { pdb_destroy(s->batch); s->batch = nullptr; } s->track.clear(); s->trackName.clear(); }

// ---- car (:219-365) ----
int addCar(int simId, const std::string& modelName) {
    logf("[PY] addCar simId=%d modelName=%s", simId, modelName.c_str());
    Sim* s = getSim(simId);
    if (!s) return -1;
    if (s->track.empty()) { logf("EXCEPTION: addCar needs a loaded track"); return -1; }
    if (s->hasCar) {   // another car of the same simulator: the next lane of its world
        if (s->numCars() >= s->maxCars) { logf("EXCEPTION: addCar: the simulator is full (cfg/sim.ini MAX_CARS = %d)", s->maxCars); return -1; }
        if (modelName != s->model) { logf("EXCEPTION: addCar: the cars of one simulator share a model in this build (%s)", s->model.c_str()); return -1; }
        auto e = std::make_unique<ExtraCar>();
        if (pdb_build_car_model(s->base.c_str(), modelName.c_str(), &e->P) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
        if (pdb_initial_state(&e->P, s->track.data(), &e->S) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
        e->S.randState = (int32_t)g_seed;
        s->pullState();
        if (s->batch) { pdb_destroy(s->batch); s->batch = nullptr; }
        s->more.push_back(std::move(e));
        {   // the wakes: the cars there keep theirs, the new car has none yet (SlipStream.h: length 0)
            const int n = s->numCars(), old = n - 1;
            std::vector<pdb_slip_state> sl(2 * (size_t)n);
            memset(sl.data(), 0, sizeof(pdb_slip_state) * sl.size());
            if (s->slips.size() == 2 * (size_t)old) for (int k = 0; k < 2; ++k) for (int c = 0; c < old; ++c) sl[(size_t)k * n + c] = s->slips[(size_t)k * old + c];
            s->slips.swap(sl);
        }
        return s->numCars() - 1;
    }
    if (pdb_build_car_model(s->base.c_str(), modelName.c_str(), &s->P) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
    if (pdb_initial_state(&s->P, s->track.data(), &s->S) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
    s->S.randState = (int32_t)g_seed;
    s->model = modelName; s->hasCar = true;
    return 0;
}
void removeCar(int simId, int carId) {
    Sim* s = getCarSim(simId, carId);
    if (!s) return;
    s->pullState();
    if (s->batch) { pdb_destroy(s->batch); s->batch = nullptr; }
    if (carId == 0) {   // the next car, if any, becomes the simulator's first
        if (s->more.empty()) s->hasCar = false;
        else { ExtraCar& e = *s->more[0]; s->P = e.P; s->S = e.S; memcpy(s->contacts, e.contacts, sizeof(s->contacts)); s->controls = e.controls; s->state = e.state; s->more.erase(s->more.begin()); }
    } else s->more.erase(s->more.begin() + (carId - 1));
    s->slips.clear();   // (the wakes start afresh)
}
void teleportCarToLocation(int simId, int carId, float x, float y, float z) { if (Sim* s = getCarSim(simId, carId)) s->teleportLocation(carId, x, y, z); }
void teleportCarToPits(int simId, int carId, int pitId) { if (Sim* s = getCarSim(simId, carId)) s->teleportPit(carId, pitId); }
void teleportCarToSpline(int simId, int carId, float d) { if (Sim* s = getCarSim(simId, carId)) s->teleportSpline(carId, d); }
void teleportCarByMode(int simId, int carId, int mode) { if (Sim* s = getCarSim(simId, carId)) s->teleportByMode(carId, mode); }
void setCarAutoTeleport(int simId, int carId, bool collision, bool badLoc, int mode) {
    if (Sim* s = getCarSim(simId, carId)) { if (pdb_set_auto_teleport(&s->carP(carId), collision, badLoc, mode) == PDB_OK) s->paramsDirty = true; }
}
void setCarControls(int simId, int carId, bool smooth, const CarControls& controls) {
    Sim* s = getCarSim(simId, carId);
    if (!s) return;
    s->carControls(carId) = controls;
    if ((s->carP(carId).smoothSteer != 0) != smooth) { s->carP(carId).smoothSteer = smooth ? 1 : 0; s->paramsDirty = true; }
}
void setCarAssists(int simId, int carId, bool autoClutch, bool autoShift, bool autoBlip) {
    if (Sim* s = getCarSim(simId, carId)) { pdb_set_assists(&s->carP(carId), autoClutch, autoShift, autoBlip, s->carP(carId).smoothSteer); s->paramsDirty = true; }
}
void getCarState(int simId, int carId, CarState& state) { if (Sim* s = getCarSim(simId, carId)) state = s->carState(carId); }
void setCarTune(int simId, int carId, const std::string& name, float value) {
    if (Sim* s = getCarSim(simId, carId)) {
        if (pdb_set_car_tune(&s->carP(carId), s->base.c_str(), s->model.c_str(), name.c_str(), value, 0) != PDB_OK) logf("EXCEPTION: %s", pdb_last_error());
        s->paramsDirty = true;
    }
}
void setCarRawTune(int simId, int carId, const std::string& name, float value) {
    if (Sim* s = getCarSim(simId, carId)) {
        if (pdb_set_car_tune(&s->carP(carId), s->base.c_str(), s->model.c_str(), name.c_str(), value, 1) != PDB_OK) logf("EXCEPTION: %s", pdb_last_error());
        s->paramsDirty = true;
    }
}
void setScoringVar(int simId, int carId, const std::string& name, float w) {
    if (Sim* s = getCarSim(simId, carId)) { pdb_set_scoring_var(&s->carP(carId), name.c_str(), w); s->paramsDirty = true; }
}
float getScoringVar(int simId, int carId, const std::string& name) { Sim* s = getCarSim(simId, carId); return s ? pdb_get_scoring_var(&s->carP(carId), name.c_str()) : 0.0f; }

// ---- vectorised extension: one lane of a batch takes the setup and the reward weights of a simulator (the reference's setCarTune / setScoringVar
//      are per simulator, i.e. per env: configure a simulator with them, then hand its values to the lanes that should drive that setup) ----
bool setBatchLaneTune(int batchId, int lane, int simId);
bool setBatchLaneSetup(int batchId, int lane, int simId);
// ---- vectorised extension: N lanes configured like simulator simId ----
int createBatch(int simId, int nCars, int device) {
    Sim* s = getSim(simId);
    if (!s || !s->hasCar || nCars <= 0) { logf("EXCEPTION: createBatch needs a simulator with a track and a car"); return -1; }
    if (s->numCars() > 1) { logf("EXCEPTION: createBatch widens a ONE-car simulator into lanes (a multi-car simulator steps its own world through stepSimulator)"); return -1; }
    auto B = std::make_shared<Batch>();
    B->b = pdb_create(device, nCars, &s->P, s->track.data(), s->track.size(), PDB_ACTION_ENV);
    if (!B->b) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
    s->pullState();
    if (pdb_set_state_all(B->b, &s->S) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
    if (s->S.numContacts > 0) {   // a car that is touching something: every lane starts with its contact joints too
        std::vector<pdb_contact> all((size_t)nCars * PDB_MAX_CONTACTS);
        for (int i = 0; i < nCars; ++i) memcpy(&all[(size_t)i * PDB_MAX_CONTACTS], s->contacts, sizeof(s->contacts));
        if (pdb_set_contacts(B->b, 0, nCars, all.data()) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return -1; }
    }
    B->n = nCars; B->out.resize(nCars);
    std::lock_guard<std::mutex> g(g_lock);
    B->id = g_uniqBatchId++;
    g_batches.insert({B->id, B});
    return B->id;
}
void destroyBatch(int id) { std::lock_guard<std::mutex> g(g_lock); g_batches.erase(id); }
bool setBatchLaneTune(int batchId, int lane, int simId) {   // simId < 0: the lane back to the batch's own block
    Batch* B = getBatch(batchId);
    if (!B || lane < 0 || lane >= B->n) return false;
    pdb_lane_tune row; memset(&row, 0, sizeof(row));
    if (simId >= 0) {
        Sim* s = getSim(simId);
        if (!s || !s->hasCar || pdb_lane_tune_from_params(&s->P, &row) != PDB_OK) return false;
    }
    if (pdb_set_lane_tunes(B->b, lane, 1, simId >= 0 ? &row : nullptr) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return false; }
    return true;
}
// the rest of simulator simId's setup for one lane (every other SetupManager tune: pdb_set_lane_setups; the batch then steps with the kernel pair compiled for the table);
// simId < 0: the lane back to its own block's values
bool setBatchLaneSetup(int batchId, int lane, int simId) {
    Batch* B = getBatch(batchId);
    if (!B || lane < 0 || lane >= B->n) return false;
    pdb_lane_setup row; memset(&row, 0, sizeof(row));
    if (simId >= 0) {
        Sim* s = getSim(simId);
        if (!s || !s->hasCar || pdb_lane_setup_from_params(&s->P, &row) != PDB_OK) return false;
    }
    if (pdb_set_lane_setups(B->b, lane, 1, simId >= 0 ? &row : nullptr) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return false; }
    return true;
}
// actions [N,2] float32 (a0 = steer, a1 -> gas like projectd_env.py:159-160) -> [N,26] float32: obs[24], reward, flags (bit-cast int32)
py::array_t<float> stepBatch(int id, py::array_t<float, py::array::c_style | py::array::forcecast> actions, double dt) {
    Batch* B = getBatch(id);
    if (!B || actions.size() != (py::ssize_t)B->n * 2) return py::array_t<float>();
    if (pdb_step_host(B->b, actions.data(), (float)dt, B->out.data()) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return py::array_t<float>(); }
    py::array_t<float> r({(py::ssize_t)B->n, (py::ssize_t)26});
    memcpy(r.mutable_data(), B->out.data(), sizeof(pdb_step_out) * (size_t)B->n);
    return r;
}
// one tick of the lanes whose hold byte is zero; the others sit it out (pdb_step_host_held).  Returns every lane's row: the held lanes' as they were
py::array_t<float> stepBatchHeld(int id, py::array_t<float, py::array::c_style | py::array::forcecast> actions, py::array_t<uint8_t, py::array::c_style | py::array::forcecast> hold, double dt) {
    Batch* B = getBatch(id);
    if (!B || actions.size() != (py::ssize_t)B->n * 2 || hold.size() != (py::ssize_t)B->n) return py::array_t<float>();
    if (pdb_step_host_held(B->b, actions.data(), (float)dt, hold.data(), B->out.data()) != PDB_OK) { logf("EXCEPTION: %s", pdb_last_error()); return py::array_t<float>(); }
    py::array_t<float> r({(py::ssize_t)B->n, (py::ssize_t)26});
    memcpy(r.mutable_data(), B->out.data(), sizeof(pdb_step_out) * (size_t)B->n);
    return r;
}
void resetBatch(int id, py::object mask, int mode) {   // teleportCarByMode(mode) for the masked lanes
    Batch* B = getBatch(id);
    if (!B) return;
    if (mask.is_none()) { pdb_reset_mode(B->b, nullptr, mode); return; }
    auto m = py::array_t<uint8_t, py::array::c_style | py::array::forcecast>::ensure(mask);
    if (m && m.size() == B->n) pdb_reset_mode(B->b, m.data(), mode);
}
// the env's reward / termination / reset rules evaluated inside the tick (pdb_set_env): one stepBatch = one VecEnv.step()
void setBatchEnv(int id, bool enabled, bool termHit, bool termOff, bool termStuck, double hitPenalty, double offPenalty, double stuckPenalty, double lowReward,
                 bool teleportOnReset, int teleportMode) {
    Batch* B = getBatch(id);
    if (!B) return;
    pdb_env_config c{};
    c.enabled = enabled; c.terminate_on_hit = termHit; c.terminate_off_track = termOff; c.terminate_when_stuck = termStuck;
    c.hit_penalty = hitPenalty; c.off_track_penalty = offPenalty; c.stuck_penalty = stuckPenalty; c.low_reward = lowReward;
    c.teleport_on_reset = teleportOnReset; c.teleport_mode = teleportMode;
    if (pdb_set_env(B->b, &c) != PDB_OK) logf("EXCEPTION: %s", pdb_last_error());
}
// env mode: the masked lanes were teleported by resetBatch; their next tick is the episode's reset tick (zero action, reward and
// termination discarded, sums cleared) -- envPending = 2: reset tick, pose already set
void markBatchResetTick(int id, py::array_t<uint8_t, py::array::c_style | py::array::forcecast> mask) {
    Batch* B = getBatch(id);
    if (!B || mask.size() != B->n) return;
    std::vector<pdb_dyn_state> st((size_t)B->n);
    if (pdb_get_state(B->b, 0, B->n, st.data()) != PDB_OK) return;
    for (int i = 0; i < B->n; ++i) if (mask.data()[i]) st[(size_t)i].envPending = 2;
    pdb_set_state(B->b, 0, B->n, st.data());
}
void clearBatchEpisodes(int id) {   // a fresh episode in every lane: the env-mode sums of the records
    Batch* B = getBatch(id);
    if (!B) return;
    std::vector<pdb_dyn_state> st((size_t)B->n);
    if (pdb_get_state(B->b, 0, B->n, st.data()) != PDB_OK) return;
    for (auto& s : st) { s.envTotalReward = 0.0; s.envPending = 0; s.envStepId = 0; }
    pdb_set_state(B->b, 0, B->n, st.data());
}
void setBatchStuckTimeout(int id, double seconds) { if (Batch* B = getBatch(id)) pdb_set_stuck_timeout(B->b, seconds); }
void setBatchSeeds(int id, py::array_t<uint32_t, py::array::c_style | py::array::forcecast> seeds) {   // one setSeed per lane
    Batch* B = getBatch(id);
    if (B && seeds.size() == B->n) pdb_set_seed(B->b, seeds.data());
}
void getBatchCarState(int id, int lane, CarState& state) {
    Batch* B = getBatch(id);
    if (!B || lane < 0 || lane >= B->n) return;
    pdb_car_state cs;
    if (pdb_get_car_state(B->b, lane, 1, &cs) == PDB_OK) { memcpy(&state, &cs, sizeof(cs)); state.carId = lane; }
}

// ---- playground / window (:367-511): rendering is out of scope, the calls are accepted ----
void launchPlaygroundInOwnThread(const std::string&) {}
void initPlayground(const std::string&) {}
void shutPlayground() {}
void shutAll() { destroyAllSimulators(); }
void tickPlayground() {}
bool isPlaygroundInitialized() { return false; }
bool isPlaygroundExited() { return false; }
void moveWindow(int, int) {}
void resizeWindow(int, int) {}
void setRenderHz(int, bool) {}
void setActiveSimulator(int, bool) {}
void setActiveCar(int, bool, bool) {}
int getActiveSimulator() { return -1; }
int getActiveCar() { return -1; }

}  // namespace

PYBIND11_MODULE(PyProjectD, m) {
    m.doc() = "PyProjectD (pdbatch: MI355X batched stepper behind the reference API)";
    py::class_<vec3f>(m, "vec3f").def(py::init<>()).def_readwrite("x", &vec3f::x).def_readwrite("y", &vec3f::y).def_readwrite("z", &vec3f::z);
    py::class_<mat44f>(m, "mat44f").def(py::init<>())
        .def_readwrite("M11", &mat44f::M11).def_readwrite("M12", &mat44f::M12).def_readwrite("M13", &mat44f::M13).def_readwrite("M14", &mat44f::M14)
        .def_readwrite("M21", &mat44f::M21).def_readwrite("M22", &mat44f::M22).def_readwrite("M23", &mat44f::M23).def_readwrite("M24", &mat44f::M24)
        .def_readwrite("M31", &mat44f::M31).def_readwrite("M32", &mat44f::M32).def_readwrite("M33", &mat44f::M33).def_readwrite("M34", &mat44f::M34)
        .def_readwrite("M41", &mat44f::M41).def_readwrite("M42", &mat44f::M42).def_readwrite("M43", &mat44f::M43).def_readwrite("M44", &mat44f::M44);
    py::class_<CarControls>(m, "CarControls").def(py::init<>())
        .def_readwrite("steer", &CarControls::steer).def_readwrite("clutch", &CarControls::clutch).def_readwrite("brake", &CarControls::brake)
        .def_readwrite("handBrake", &CarControls::handBrake).def_readwrite("gas", &CarControls::gas)
        .def_readwrite("isShifterSupported", &CarControls::isShifterSupported).def_readwrite("requestedGearIndex", &CarControls::requestedGearIndex)
        .def_readwrite("gearUp", &CarControls::gearUp).def_readwrite("gearDn", &CarControls::gearDn);
    py::class_<CarState>(m, "CarState").def(py::init<>())
        .def_readonly("carId", &CarState::carId).def_readonly("simId", &CarState::simId).def_readonly("timestamp", &CarState::timestamp)
        .def_readonly("controls", &CarState::controls)
        .def_readonly("collisionFlag", &CarState::collisionFlag).def_readonly("outOfTrackFlag", &CarState::outOfTrackFlag)
        .def_readonly("trackPointId", &CarState::trackPointId).def_readonly("lastTrackPointTimestamp", &CarState::lastTrackPointTimestamp)
        .def_readonly("trackLocation", &CarState::trackLocation).def_readonly("bodyVsTrack", &CarState::bodyVsTrack)
        .def_readonly("velocityVsTrack", &CarState::velocityVsTrack)
        .def_readonly("engineRPM", &CarState::engineRPM).def_readonly("speedMS", &CarState::speedMS).def_readonly("gear", &CarState::gear)
        .def_readonly("gearGrinding", &CarState::gearGrinding)
        .def_readonly("bodyMatrix", &CarState::bodyMatrix).def_readonly("bodyPos", &CarState::bodyPos).def_readonly("bodyEuler", &CarState::bodyEuler)
        .def_readonly("accG", &CarState::accG).def_readonly("velocity", &CarState::velocity).def_readonly("localVelocity", &CarState::localVelocity)
        .def_readonly("angularVelocity", &CarState::angularVelocity).def_readonly("localAngularVelocity", &CarState::localAngularVelocity)
        .def_readonly("hubMatrix", &CarState::hubMatrix).def_readonly("tyreContacts", &CarState::tyreContacts)
        .def_readonly("tyreLoad", &CarState::tyreLoad).def_readonly("tyreAngularSpeed", &CarState::tyreAngularSpeed)
        .def_readonly("tyreSlipRatio", &CarState::tyreSlipRatio).def_readonly("tyreNdSlip", &CarState::tyreNdSlip)
        .def_readonly("probes", &CarState::probes).def_readonly("lookAhead", &CarState::lookAhead)
        .def_readonly("stepReward", &CarState::stepReward).def_readonly("totalReward", &CarState::totalReward);

    m.def("setSeed", &setSeed, "");
    m.def("setLogFile", &setLogFile, "", py::arg("filename"), py::arg("overwrite") = true);
    m.def("clearLogFile", &clearLogFile, "");
    m.def("writeLog", &writeLog, "");
    m.def("createSimulator", &createSimulator, "");
    m.def("destroySimulator", &destroySimulator, "");
    m.def("stepSimulator", &stepSimulator, "", py::arg("simId"), py::arg("dt") = 1.0 / 333.0);
    m.def("loadTrack", &loadTrack, "");
    m.def("unloadTrack", &unloadTrack, "");
    m.def("addCar", &addCar, "");
    m.def("removeCar", &removeCar, "");
    m.def("teleportCarToLocation", &teleportCarToLocation, "");
    m.def("teleportCarToPits", &teleportCarToPits, "");
    m.def("teleportCarToSpline", &teleportCarToSpline, "");
    m.def("teleportCarByMode", &teleportCarByMode, "");
    m.def("setCarAutoTeleport", &setCarAutoTeleport, "", py::arg("simId"), py::arg("carId"), py::arg("collision"), py::arg("badLoc"), py::arg("teleportMode") = 0);
    m.def("setCarControls", &setCarControls, "");
    m.def("setCarAssists", &setCarAssists, "");
    m.def("getCarState", &getCarState, "");
    m.def("setCarRawTune", &setCarRawTune, "");
    m.def("setCarTune", &setCarTune, "");
    m.def("setScoringVar", &setScoringVar, "");
    m.def("getScoringVar", &getScoringVar, "");
    m.def("launchPlaygroundInOwnThread", &launchPlaygroundInOwnThread, "");
    m.def("initPlayground", &initPlayground, "");
    m.def("shutPlayground", &shutPlayground, "");
    m.def("shutAll", &shutAll, "");
    m.def("tickPlayground", &tickPlayground, "");
    m.def("isPlaygroundInitialized", &isPlaygroundInitialized, "");
    m.def("isPlaygroundExited", &isPlaygroundExited, "");
    m.def("moveWindow", &moveWindow, "");
    m.def("resizeWindow", &resizeWindow, "");
    m.def("setRenderHz", &setRenderHz, "");
    m.def("setActiveSimulator", &setActiveSimulator, "");
    m.def("setActiveCar", &setActiveCar, "");
    m.def("getActiveSimulator", &getActiveSimulator, "");
    m.def("getActiveCar", &getActiveCar, "");
    // vectorised extension (not in the reference): one configured simulator widened to N lanes
    m.def("createBatch", &createBatch, "", py::arg("simId"), py::arg("nCars"), py::arg("device") = 0);
    m.def("destroyBatch", &destroyBatch, "");
    m.def("setBatchLaneTune", &setBatchLaneTune, "", py::arg("batchId"), py::arg("lane"), py::arg("simId"));
    m.def("setBatchLaneSetup", &setBatchLaneSetup, "", py::arg("batchId"), py::arg("lane"), py::arg("simId"));
    m.def("stepBatch", &stepBatch, "", py::arg("batchId"), py::arg("actions"), py::arg("dt") = 1.0 / 333.0);
    m.def("stepBatchHeld", &stepBatchHeld, "", py::arg("batchId"), py::arg("actions"), py::arg("hold"), py::arg("dt") = 1.0 / 333.0);
    m.def("resetBatch", &resetBatch, "", py::arg("batchId"), py::arg("mask") = py::none(), py::arg("mode") = 0);
    m.def("setBatchStuckTimeout", &setBatchStuckTimeout, "");
    m.def("setBatchEnv", &setBatchEnv, "");
    m.def("clearBatchEpisodes", &clearBatchEpisodes, "");
    m.def("markBatchResetTick", &markBatchResetTick, "");
    m.def("setBatchSeeds", &setBatchSeeds, "");
    m.def("getBatchCarState", &getBatchCarState, "");
}
