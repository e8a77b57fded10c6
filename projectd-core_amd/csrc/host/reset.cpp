// pdbatch host side: construction-time state and the teleport/reset edits of the reference
//   Car::reset                      Car/Car.cpp:385-410
//   Car::forcePosition/forceRotation/teleport/teleportToPits/teleportToSpline   Car/Car.cpp:1240-1340
//   SuspensionStrut::attach/setPositions/stop           Car/SuspensionStrut.cpp:146-228
//   SuspensionAxle::attach/stop                         Car/SuspensionAxle.cpp:105-118
//   Tyre::reset, TyreThermalModel::reset                Car/Tyre.cpp:393-425, Car/TyreThermalModel.cpp:194-208
//   Drivetrain::reset/setCurrentGear, Engine::reset     Car/Drivetrain.cpp:154-207, Car/Engine.cpp:160-168
//   ScoringSystem::reset                                Car/ScoringSystem.cpp:120-127
// Episode resets are rare (once per episode) and sequential; they edit the per-car record on the
// host and the batch uploads it -- the per-tick path never runs on the host.
#include "model.hpp"
#include "reset_core.hpp"

namespace pdb {

struct HostRayDown {
    const TrackView& tv;
    bool operator()(const float* o, float& hitY) const {
        const float d[3] = {0, -1, 0};
        const RayHitH hit = rayCastTrack(tv, o, d, 1000.0f);
        if (hit.has) hitY = hit.pos[1];
        return hit.has;
    }
};

void teleportToSpline(const pdb_car_params& P, const TrackView& tv, float distanceNorm, pdb_dyn_state& S) {
    teleportToSplineT(P, tv.h->numFat, tv.fat, HostRayDown{tv}, distanceNorm, S);
}
void teleportToPit(const pdb_car_params& P, const TrackView& tv, int pitId, pdb_dyn_state& S) {
    teleportToPitT(P, tv.h->numPits, tv.pits, HostRayDown{tv}, pitId, S);
}
void teleportToLocation(const pdb_car_params& P, const TrackView& tv, const float* pos, pdb_dyn_state& S) {
    forcePositionT(P, HostRayDown{tv}, pos, S);
}
void teleportByMode(const pdb_car_params& P, const TrackView& tv, int mode, pdb_dyn_state& S) {
    teleportByModeT(P, tv.h->numFat, tv.fat, HostRayDown{tv}, mode, S);
}

void initialState(const pdb_car_params& P, const TrackView& tv, pdb_dyn_state& S) {
    memset(&S, 0, sizeof(S));
    for (int i = 0; i < PDB_MAX_BODIES; ++i) {
        S.body[i].q[0] = 1; S.body[i].R[0] = 1; S.body[i].R[4] = 1; S.body[i].R[8] = 1;
    }
    // Car::init at the identity pose
    memcpy(S.body[PDB_BODY_TANK].pos, P.fuelTankPos, 12);
    attachAll(P, S);
    for (int i = 0; i < 4; ++i) {
        pdb_tyre_state& t = S.tyre[i];
        t.pressureDynamic = P.tyre[i].pressureRef;   // Tyre::setCompound (Tyre.cpp:355-356)
        t.thermalMultD = 1.0f;                        // TyreThermalModel.h:51
        t.inputT0 = P.ambientTemperature;             // TyreThermalModel::buildTyre (TyreThermalModel.cpp:43)
        tyreReset(P, t);
    }
    S.fuel = P.fuel;
    S.fuelPressure = 1.0f; S.lifeLeft = 1000.0f;
    S.currentGear = 1;                               // Drivetrain::init -> setCurrentGear(1, true)
    S.gearReqTimeout = 200; S.gearReqRequestedGear = -1; S.lastRatio = -1.0;
    S.validShiftRPMWindow = P.validShiftRPMWindow;
    S.acSeqIsDone = 1;
    S.locClutch = 1.0f;                              // Drivetrain.h:137
    S.pointCachePos[0] = 0; S.pointCachePos[1] = -10000.0f; S.pointCachePos[2] = 0;   // Track.cpp:203
    S.randState = 1;                                 // the C runtime's default seed (srand(1))
    teleportToSpline(P, tv, 0.0f, S);
}

}  // namespace pdb
