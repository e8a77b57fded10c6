// pdbatch: construction-time state and the teleport / reset edits of the reference, shared by the host library and the HIP
// kernels (one source, compiled by g++ and by hipcc):
//   Car::reset                      Car/Car.cpp:385-410
//   Car::forcePosition/forceRotation/teleportToSpline/teleportByMode   Car/Car.cpp:1240-1340
//   SuspensionStrut::attach/setPositions/stop           Car/SuspensionStrut.cpp:146-228
//   SuspensionAxle::attach/stop                         Car/SuspensionAxle.cpp:105-118
//   Tyre::reset, TyreThermalModel::reset                Car/Tyre.cpp:393-425, Car/TyreThermalModel.cpp:194-208
//   Drivetrain::reset/setCurrentGear, Engine::reset     Car/Drivetrain.cpp:154-207, Car/Engine.cpp:160-168
//   ScoringSystem::reset                                Car/ScoringSystem.cpp:120-127
#pragma once
#include "rbmath.hpp"
#include <cstring>
#include <cmath>

namespace pdb {

PDB_HD inline void toHBody(const pdb_body_state& s, HBody& b) {
    memcpy(b.pos, s.pos, 12); memcpy(b.q, s.q, 16); memcpy(b.R, s.R, 36);
}
PDB_HD inline void fromHBody(const HBody& b, pdb_body_state& s) {
    memcpy(s.pos, b.pos, 12); memcpy(s.q, b.q, 16); memcpy(s.R, b.R, 36);
}
PDB_HD inline void stopBody(pdb_body_state& s) { for (int k = 0; k < 3; ++k) { s.lvel[k] = 0; s.avel[k] = 0; } }
PDB_HD inline float v3len(const float* a) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
PDB_HD inline void v3norm(float* a) { const float l = v3len(a); if (l != 0.0f) { const float s = 1.0f / l; a[0] *= s; a[1] *= s; a[2] *= s; } }

// ISuspension::attach() for every wheel (joints already exist => setPositions only)
PDB_HD inline void attachAll(const pdb_car_params& P, pdb_dyn_state& S) {
    HBody body; toHBody(S.body[PDB_BODY_CHASSIS], body);
    float Mb[9]; hWorldMatrix3(body, Mb);
    for (int i = 0; i < 4; ++i) {
        const pdb_susp& su = P.susp[i];
        if (su.type == PDB_SUSP_STRUT) {
            HBody hub, strut;
            toHBody(S.body[su.hubBody], hub); toHBody(S.body[su.strutBody], strut);
            float vPos[3]; hLocalToWorld(body, su.basePosition, vPos);
            hSetRotationM(hub, Mb);
            memcpy(hub.pos, vPos, 12);
            float vCarStrut[3], vTyreStrut[3], vNorm[3];
            hLocalToWorld(body, su.carStrut, vCarStrut);
            hLocalToWorld(hub, su.tyreStrut, vTyreStrut);
            for (int k = 0; k < 3; ++k) vNorm[k] = vTyreStrut[k] - vCarStrut[k];
            v3norm(vNorm);
            const float vM3[3] = {Mb[6] * -1.0f, Mb[7] * -1.0f, Mb[8] * -1.0f};
            float vM3N[3], vM3NN[3];
            hCross(vM3N, vM3, vNorm);
            hCross(vM3NN, vM3N, vNorm); v3norm(vM3NN);
            const float Ms[9] = {vM3NN[0], vM3NN[1], vM3NN[2], -vM3N[0], -vM3N[1], -vM3N[2], -vNorm[0], -vNorm[1], -vNorm[2]};
            hSetRotationM(strut, Ms);
            for (int k = 0; k < 3; ++k) strut.pos[k] = (vNorm[k] * su.strutBodyLength) * 0.5f + vCarStrut[k];
            fromHBody(hub, S.body[su.hubBody]); fromHBody(strut, S.body[su.strutBody]);
        } else if (su.type == PDB_SUSP_DW || su.type == PDB_SUSP_ML) {   // SuspensionDW::attach (SuspensionDW.cpp:156-162), SuspensionML::attach (:95-99)
            HBody hub; toHBody(S.body[su.hubBody], hub);
            hSetRotationM(hub, Mb);
            hLocalToWorld(body, su.basePosition, hub.pos);
            fromHBody(hub, S.body[su.hubBody]);
        } else if (su.type == PDB_SUSP_AXLE && su.sideSign > 0.0f) {
            HBody axle; toHBody(S.body[su.hubBody], axle);
            hSetRotationM(axle, Mb);
            hLocalToWorld(body, su.axleBasePos, axle.pos);
            fromHBody(axle, S.body[su.hubBody]);
        }
    }
}

PDB_HD inline void tyreReset(const pdb_car_params& P, pdb_tyre_state& t) {
    t.slipAngleRAD = 0; t.slipRatio = 0; t.angularVelocity = 0; t.Fy = 0; t.Fx = 0; t.Mz = 0;
    t.isLocked = 1; t.inflation = 1; t.flatSpot = 0;
    t.dirtyLevel = 0; t.virtualKM = 0;
    t.coreTemp = P.ambientTemperature;
    for (int k = 0; k < 36; ++k) t.T[k] = P.ambientTemperature;
    t.phase = 0;
}

// Car::forceRotation(heading) (Car.cpp:1274-1308): the chassis and the tank take the matrix built from the heading, the suspensions are
// re-attached, both bodies stop
PDB_HD inline void forceRotationT(const pdb_car_params& P, const float* heading, pdb_dyn_state& S) {
    const float ihed[3] = {heading[0] * -1.0f, heading[1] * -1.0f, heading[2] * -1.0f};
    const float vM13 = ihed[0], vM11 = -ihed[2], vM12 = 0;
    const float v6 = sqrtf((vM12 * vM12) + (vM11 * vM11) + (vM13 * vM13));
    const float s = 1.0f / v6;
    const float M[9] = {vM11 * s, vM12 * s, vM13 * s, 0, 1, 0, -ihed[0], -ihed[1], -ihed[2]};
    HBody b; toHBody(S.body[PDB_BODY_CHASSIS], b);
    hSetRotationM(b, M); fromHBody(b, S.body[PDB_BODY_CHASSIS]);
    HBody t; toHBody(S.body[PDB_BODY_TANK], t);
    hSetRotationM(t, M); fromHBody(t, S.body[PDB_BODY_TANK]);
    attachAll(P, S);
    stopBody(S.body[PDB_BODY_CHASSIS]); stopBody(S.body[PDB_BODY_TANK]);
}

// Car::forcePosition(pos, offsetY = 0) (Car.cpp:1240-1272).  rayDown(origin, hitY) = the engine's ray cast straight down from `origin`
// over 1000 m (Car.cpp:1243): true + the hit's y
template <class RayDown>
PDB_HD inline void forcePositionT(const pdb_car_params& P, RayDown rayDown, const float* pos, pdb_dyn_state& S) {
    float bodyPos[3] = {pos[0], pos[1], pos[2]};
    const float o[3] = {pos[0] + 0.0f, pos[1] + 10.0f, pos[2] + 0.0f};
    float hitY;
    if (rayDown(o, hitY)) bodyPos[1] = hitY;
    bodyPos[1] += (P.baseCarHeight + 0.0f + 0.01f);
    // Car::reset()
    S.waterT = 60.0f;
    for (int i = 0; i < 4; ++i) S.brakeDiscT[i] = P.ambientTemperature;   // BrakeSystem::reset (BrakeSystem.cpp:73-80)
    S.fuel = P.fuel;
    S.collisionFlag = 0; S.oldCollisionFlag = 0; S.outOfTrackFlag = 0;
    S.lastTrackPointTimestamp = (float)S.physicsTime;
    S.nearestTrackPointId = 0; S.oldTrackPointId = 0; S.splinePointId = 0;
    S.trackLocation = 0; S.oldTrackLocation = 0;
    for (int i = 0; i < 5; ++i) S.damageZoneLevel[i] = 0;   // Car.cpp:403-407 (the simulator's collision frame counter runs on)
    S.damageChanged = 0;
    S.totalReward = 0; S.stepReward = 0; S.oldPointId = 0; S.oldSplinePointId = 0;   // ScoringSystem::reset
    stopBody(S.body[PDB_BODY_CHASSIS]);
    memcpy(S.body[PDB_BODY_CHASSIS].pos, bodyPos, 12);
    HBody b; toHBody(S.body[PDB_BODY_CHASSIS], b);
    hLocalToWorld(b, P.fuelTankPos, S.body[PDB_BODY_TANK].pos);
    for (int i = 0; i < 4; ++i) stopBody(S.body[P.susp[i].hubBody]);   // ISuspension::stop(): hub / axle only
    attachAll(P, S);
    // Drivetrain::reset + Engine::reset
    S.clutchOpenState = 1; S.rootVelocity = 0; S.engineVel = 0; S.outShaftLVel = 0; S.outShaftRVel = 0; S.driveVel = 0;
    S.gearReqRequest = 0; S.validShiftRPMWindow = P.validShiftRPMWindow; S.lifeLeft = 1000.0f;
    for (int i = 0; i < PDB_MAX_TURBOS; ++i) S.turboRotation[i] = 0.0f;   // Engine::reset -> Turbo::reset (Engine.cpp:160-166, Turbo.cpp:42-45)
    for (int i = 0; i < 4; ++i) tyreReset(P, S.tyre[i]);
    S.isGearGrinding = 0; S.currentGear = 1;   // setCurrentGear(1, true)
    stopBody(S.body[PDB_BODY_CHASSIS]); stopBody(S.body[PDB_BODY_TANK]);
}

// Car::teleport(m) (Car.cpp:1310-1314) = forceRotation(M31..M33) then forcePosition(M41..M43); Car::teleportToPits(pitId) (Car.cpp:1316-1323) =
// teleport(track->pits[pitId]), nothing for an id outside the list.  pits = float[numPits][16] (the blob's pit matrices, row by row)
template <class RayDown>
PDB_HD inline void teleportToPitT(const pdb_car_params& P, int numPits, const float* pits, RayDown rayDown, int pitId, pdb_dyn_state& S) {
    if (pitId < 0 || pitId >= numPits) return;
    const float* m = pits + 16 * pitId;
    forceRotationT(P, m + 8, S);
    forcePositionT(P, rayDown, m + 12, S);
}

// Car::teleportToSpline (Car.cpp:1325-1340).  fat = the track's fat points (15 floats each)
template <class RayDown>
PDB_HD inline void teleportToSplineT(const pdb_car_params& P, int numFat, const float* fat, RayDown rayDown, float distanceNorm, pdb_dyn_state& S) {
    const int n = numFat;
    if (!n) return;
    // Track::getPointIdAtDistance (Track.cpp:556-570)
    if (distanceNorm < 0.0f) distanceNorm += 1.0f; else if (distanceNorm > 1.0f) distanceNorm -= 1.0f;
    const float dn = distanceNorm < 0.0f ? 0.0f : (distanceNorm > 1.0f ? 1.0f : distanceNorm);
    const size_t pointId = (size_t)(dn * (float)(n - 1));
    if (pointId >= (size_t)n) return;
    const float* pt = fat + 15 * pointId;
    forceRotationT(P, pt + 12, S);        // forwardDir
    forcePositionT(P, rayDown, pt + 9, S);   // center
}


// rand() of the reference's C runtime (Core/Math.h:49-52 randR; the reference is built with MSVC: holdrand = holdrand * 214013 +
// 2531011, rand() = (holdrand >> 16) & 0x7fff, RAND_MAX = 32767), one generator per car -- the reference env runs one simulator
// per process, each with its own CRT state (srand(1) by default, setSeed = srand: PyProjectD.cpp:50-53)
PDB_HD inline float randR01(uint32_t& holdrand) {
    holdrand = holdrand * 214013u + 2531011u;
    const int r = (int)((holdrand >> 16) & 0x7fffu);
    const float f = (float)r / (float)32767;
    return 0.0f + f * (1.0f - 0.0f);
}
// Car::teleportByMode (Car.cpp:1320-1336): 0 Start, 1 Nearest (the car's trackLocation), 2 Random
template <class RayDown>
PDB_HD inline void teleportByModeT(const pdb_car_params& P, int numFat, const float* fat, RayDown rayDown, int mode, pdb_dyn_state& S) {
    float d = 0.0f;
    if (mode == 1) d = S.trackLocation;
    else if (mode == 2) { uint32_t h = (uint32_t)S.randState; d = randR01(h); S.randState = (int32_t)h; }
    else if (mode != 0) return;
    teleportToSplineT(P, numFat, fat, rayDown, d, S);
}

}  // namespace pdb
