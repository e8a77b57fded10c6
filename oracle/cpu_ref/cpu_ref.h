// ORACLE / TEST INFRASTRUCTURE -- not product code.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may link or run this.
//
// cpu_ref: scalar, one-car-at-a-time CPU restatement of the reference's per-tick vehicle step
//   Simulator::step -> Car::step -> stepComponents -> IPhysicsEngine::step -> Car::postStep
// (reference src/ProjectD/Sim/Simulator.cpp:168-237, Car/Car.cpp:414-865), arithmetic in the
// reference's own types (float, double where the reference holds double).  Each function cites the
// reference lines it follows.  Input data (pdb_car_params, the track blob, pdb_dyn_state) use the
// product's documented formats from include/pdb_types.h.
//
// Pinning: every quantity above the IPhysicsEngine seam is compared tick-by-tick against golden
// trajectories produced by the reference's own translation units (oracle/refharness ->
// tests/golden/*.npz).  The rigid-body solve (oracle/rb) is PARITY UNPINNED: ODE is absent.
#pragma once
#include "pdb_types.h"
#include "../rb/pdrb.h"
#include "../rb/pdcollide.h"
#include "../probe.h"
#include <vector>
#include <cstdint>

namespace cpuref {

struct TrackData {
    const pdb_track_header* h = nullptr;
    const pdb_surface* surfaces = nullptr;
    const float* tris = nullptr;
    const float* fat = nullptr;
    const float* fatDist = nullptr;
    const float* nodes = nullptr;
    const float* nodeDist = nullptr;
    void bind(const uint8_t* blob);
};

// per-tick scratch that is not persistent state but is observable (probe / CarState)
struct TyreScratch {
    float brakeTorque = 0, handBrakeTorque = 0, feedbackTorque = 0, rollingResistence = 0, thermalInput = 0;
    float slipFactor = 0, Dx = 0, Dy = 0, depth = 0, distToGround = 0, liveRadius = 0, wearMult = 0;
    float totalHubVelocity = 0, slidingVelocityX = 0, slidingVelocityY = 0, roadVelocityX = 0;
    float travel = 0, damperSpeedMS = 0;
    int surface = -1;
    float hubMatrix[16];
};
struct WingScratch { float aoa = 0, yawAngle = 0, cd = 0, cl = 0, dragKG = 0, liftKG = 0, groundHeight = 0; };

struct Car {
    const pdb_car_params* P = nullptr;
    const TrackData* T = nullptr;
    pdb_dyn_state S;
    pdrb::World w;
    // transient
    pdb_controls controls;
    float finalSteerAngleSignal = 0;
    float accG[3] = {0, 0, 0};
    float probeHits[7];
    float lookAhead[5];
    TyreScratch ts[4];
    WingScratch ws[PDB_MAX_WINGS];
    float turboBoost = 0;
    double locClutch = 1.0, currentClutchTorque = 0, ratio = 12.0, totalTorque = 0, engOutTorque = 0;
    float gasUsage = 0;
    double stepTime = 0;   // sim->physicsTime seen by the last step (before += dt)
    int acSeqCount = 0;
    std::vector<int> nearby;
    void (*autoTeleportHook)(pdb_dyn_state*, int mode) = nullptr;   // Car::teleportByMode on a state record (the product's pdb_teleport_by_mode)
    // multi-car simulators: the car's own slipstream as its last postStep left it (Sim/SlipStream.cpp:37-47; zero until the first tick ends), the other cars' as THEIR last
    // postStep left them (handed in by whoever steps the world: every car of a Simulator steps before any postStep, Simulator.cpp:168-201), the tick's air density
    int physicsGUID = 0;   // the car's index in its simulator (Car.h physicsGUID): the first car alone switches its suspension joints' ERP by speed (Car.cpp:426)
    pdb_slip_state slip;
    std::vector<pdb_slip_state> otherSlips;
    float airDensityNow = 0;
    pdcol::ContactSet contactSet;   // the engine's contactGroupDynamic for this car (S.numContacts of them are alive)
    int contactCandidates = 0;      // diagnostic: contact points the last odd frame produced, before the PDB_MAX_CONTACTS cut

    void init(const pdb_car_params* P, const TrackData* T, const pdb_dyn_state& s0);
    void loadState(const pdb_dyn_state& s);     // pdb_dyn_state -> bodies (contactSet: setContacts)
    void setContacts(const pdb_contact* c, int n);
    void getContacts(pdb_contact* c) const;     // PDB_MAX_CONTACTS entries, the first S.numContacts alive
    void storeState();                          // bodies -> S
    void step(float steer, float gas, float dt, double dtD);   // one env tick (setCarControls + stepSimulator)
    void stepControls(const pdb_controls& c, float dt, double dtD);   // same with every CarControls field given
    void fillCarState(pdb_car_state& cs) const;
    void fillStepOut(pdb_step_out& o) const;
    void fillProbe(pdoracle::Probe& P) const;

   private:
    void carStep(float dt);
    void collisionStep();
    void postStep(float dt);
};

}  // namespace cpuref
