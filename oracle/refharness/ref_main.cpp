// ORACLE / TEST INFRASTRUCTURE (fixture generator; runs only in the build container, where
// /root/reference is mounted).  Drives the reference's own Simulator / Track / Car translation
// units (compiled in place from /root/reference/src/ProjectD, never copied) through the exact call
// sequence of pyprojectd/projectd_env.py:118-227 and records a probe vector per tick.  The
// rigid-body engine behind IPhysicsEngine is this project's pdrb (see ref_physics.cpp): everything
// above that seam -- suspension, tyre, drivetrain, engine, aero, assists, scoring, track queries --
// is the reference's code; the solver is not (ODE absent => that part is unpinned).
#include "Sim/Simulator.h"
#include "Sim/Track.h"
#include "Car/CarImpl.h"
#include "ref_physics.h"
#include "../probe.h"
#include "../scenarios.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>

using namespace D;
using pdoracle::Probe;

static void fillProbe(Probe& P, Simulator* sim, Car* car, int carIndex = 0, int numCars = 1) {
    pdrb::World* w = ref_get_world(sim->physics.get());
    P.p("time", sim->physicsTime);
    // rigid bodies in creation order (= world order): chassis, tank, [axle], then per wheel hub (+ strut body)
    const bool legacy = car->suspensions[0]->getType() == SuspensionType::Strut && car->suspensions[2]->getType() == SuspensionType::Axle;
    const char* bn[7] = {"chassis", "tank", "axle", "hub0", "strut0", "hub1", "strut1"};
    char nm[96], bname[16];
    const int nbCar = (int)w->bodies.size() / numCars;   // (cars of one model: the world holds their bodies car after car, in creation order)
    for (int i = 0; i < nbCar; ++i) {
        const pdrb::Body& b = w->bodies[carIndex * nbCar + i];
        if (!legacy) { snprintf(bname, sizeof(bname), "body%d", i); }
        const char* bni = legacy ? bn[i] : bname;
#define bn_i bni
        snprintf(nm, sizeof(nm), "%s.pos", bn_i); P.p3(nm, b.pos);
        snprintf(nm, sizeof(nm), "%s.q", bn_i); P.pn(nm, b.q, 4);
        snprintf(nm, sizeof(nm), "%s.R", bn_i); P.pn(nm, b.R, 9);
        snprintf(nm, sizeof(nm), "%s.lvel", bn_i); P.p3(nm, b.lvel);
        snprintf(nm, sizeof(nm), "%s.avel", bn_i); P.p3(nm, b.avel);
    }
#undef bn_i
    const auto& c = car->controls;
    P.p("ctrl.steer", c.steer); P.p("ctrl.clutch", c.clutch); P.p("ctrl.brake", c.brake);
    P.p("ctrl.handBrake", c.handBrake); P.p("ctrl.gas", c.gas); P.p("ctrl.gearUp", c.gearUp); P.p("ctrl.gearDn", c.gearDn);
    P.p("car.finalSteerAngleSignal", car->finalSteerAngleSignal);
    P.p("car.smoothSteerValue", car->smoothSteerValue);
    P.p3("car.accG", &car->accG.x);
    P.p("car.sleepingFrames", car->sleepingFrames);
    P.p("car.speed", car->speed.value);
    P.p3("car.lastVelocity", &car->lastVelocity.x);
    P.p("car.waterT", car->water->t);
    P.p("car.fuel", car->fuel);
    P.p("aero.airDensity", car->aeroMap->airDensity);
    for (int i = 0; i < 4; ++i) {
        Tyre* t = car->tyres[i].get();
        const TyreStatus& st = t->status;
#define TP(field, val) snprintf(nm, sizeof(nm), "tyre%d." field, i); P.p(nm, val)
        TP("angularVelocity", st.angularVelocity); TP("slipAngleRAD", st.slipAngleRAD); TP("slipRatio", st.slipRatio);
        TP("ndSlip", st.ndSlip); TP("load", st.load); TP("Fx", st.Fx); TP("Fy", st.Fy); TP("Mz", st.Mz);
        TP("isLocked", st.isLocked ? 1 : 0); TP("dirtyLevel", st.dirtyLevel); TP("flatSpot", st.flatSpot);
        TP("inflation", st.inflation); TP("pressureDynamic", st.pressureDynamic); TP("pressureStatic", st.pressureStatic);
        TP("loadedRadius", st.loadedRadius); TP("effectiveRadius", st.effectiveRadius); TP("liveRadius", st.liveRadius);
        TP("camberRAD", st.camberRAD); TP("D", st.D); TP("Dx", st.Dx); TP("Dy", st.Dy); TP("depth", st.depth);
        TP("distToGround", st.distToGround); TP("feedbackTorque", st.feedbackTorque);
        TP("rollingResistence", st.rollingResistence); TP("thermalInput", st.thermalInput);
        TP("slipFactor", st.slipFactor); TP("virtualKM", st.virtualKM); TP("wearMult", st.wearMult);
        TP("grain", st.grain); TP("blister", st.blister);
        TP("localMX", t->localMX); TP("oldAngularVelocity", t->oldAngularVelocity);
        TP("totalHubVelocity", t->totalHubVelocity); TP("slidingVelocityX", t->slidingVelocityX);
        TP("slidingVelocityY", t->slidingVelocityY); TP("roadVelocityX", t->roadVelocityX);
        TP("brakeTorque", t->inputs.brakeTorque); TP("handBrakeTorque", t->inputs.handBrakeTorque);
        snprintf(nm, sizeof(nm), "tyre%d.contactPoint", i); P.p3(nm, &t->contactPoint.x);
        snprintf(nm, sizeof(nm), "tyre%d.unmodifiedContactPoint", i); P.p3(nm, &t->unmodifiedContactPoint.x);
        snprintf(nm, sizeof(nm), "tyre%d.contactNormal", i); P.p3(nm, &t->contactNormal.x);
        TyreThermalModel* th = t->thermalModel.get();
        TP("coreTemp", th->coreTemp); TP("phase", th->phase); TP("thermalMultD", th->thermalMultD);
        TP("practicalTemp", th->practicalTemp);
        for (int k = 0; k < 36; ++k) { snprintf(nm, sizeof(nm), "tyre%d.T[%d]", i, k); P.p(nm, th->patches[k].T); }
        auto ss = car->suspensions[i]->getStatus();
        TP("susp.travel", ss.travel); TP("susp.damperSpeedMS", ss.damperSpeedMS);
#undef TP
    }
    Drivetrain* d = car->drivetrain.get();
    Engine* e = d->engineModel.get();
    P.p("dt.engine.velocity", d->engine.velocity); P.p("dt.drive.velocity", d->drive.velocity);
    P.p("dt.outShaftL.velocity", d->outShaftL.velocity); P.p("dt.outShaftR.velocity", d->outShaftR.velocity);
    P.p("dt.rootVelocity", d->rootVelocity); P.p("dt.locClutch", d->locClutch);
    P.p("dt.currentClutchTorque", d->currentClutchTorque); P.p("dt.ratio", d->ratio); P.p("dt.lastRatio", d->lastRatio);
    P.p("dt.cutOff", d->cutOff); P.p("dt.totalTorque", d->totalTorque); P.p("dt.currentGear", d->currentGear);
    P.p("dt.isGearGrinding", d->isGearGrinding ? 1 : 0); P.p("dt.clutchOpenState", d->clutchOpenState ? 1 : 0);
    P.p("dt.gearRequest.request", (int)d->gearRequest.request); P.p("dt.gearRequest.timeAccumulator", d->gearRequest.timeAccumulator);
    P.p("dt.gearRequest.timeout", d->gearRequest.timeout); P.p("dt.gearRequest.requestedGear", d->gearRequest.requestedGear);
    P.p("dt.validShiftRPMWindow", d->validShiftRPMWindow);
    P.p("eng.outTorque", e->status.outTorque); P.p("eng.limiterOn", e->limiterOn); P.p("eng.lifeLeft", e->lifeLeft);
    P.p("eng.gasUsage", e->gasUsage); P.p("eng.fuelPressure", e->fuelPressure); P.p("eng.turboBoost", e->status.turboBoost);
    P.p("ac.clutchValueSignal", car->autoClutch->clutchValueSignal);
    P.p("ac.seq.currentTime", car->autoClutch->clutchSequence.currentTime);
    P.p("ac.seq.isDone", car->autoClutch->clutchSequence.isDone ? 1 : 0);
    P.p("ac.seq.count", car->autoClutch->clutchSequence.clutchCurve.getCount());
    P.p("ab.blipStartTime", car->autoBlip->blipStartTime);
    P.p("as.gasCutoff", car->autoShift->gasCutoff);
    P.p("gc.lastGearUp", car->gearChanger->lastGearUp ? 1 : 0); P.p("gc.lastGearDn", car->gearChanger->lastGearDn ? 1 : 0);
    for (size_t i = 0; i < car->aeroMap->wings.size(); ++i) {
        const WingState& ws = car->aeroMap->wings[i]->status;
#define WP(field, val) snprintf(nm, sizeof(nm), "wing%d." field, (int)i); P.p(nm, val)
        WP("aoa", ws.aoa); WP("yawAngle", ws.yawAngle); WP("cd", ws.cd); WP("cl", ws.cl);
        WP("dragKG", ws.dragKG); WP("liftKG", ws.liftKG); WP("groundHeight", ws.groundHeight);
#undef WP
    }
    P.p("trk.nearestTrackPointId", car->nearestTrackPointId); P.p("trk.oldTrackPointId", car->oldTrackPointId);
    P.p("trk.splinePointId", car->splinePointId); P.p("trk.lastTrackPointTimestamp", car->lastTrackPointTimestamp);
    P.p("trk.trackLocation", car->trackLocation); P.p("trk.oldTrackLocation", car->oldTrackLocation);
    P.p("trk.bodyVsTrack", car->bodyVsTrack); P.p("trk.velocityVsTrack", car->velocityVsTrack);
    for (int i = 0; i < 7; ++i) { snprintf(nm, sizeof(nm), "trk.probe[%d]", i); P.p(nm, car->probeHits[i]); }
    for (int i = 0; i < 5; ++i) { snprintf(nm, sizeof(nm), "trk.lookAhead[%d]", i); P.p(nm, car->lookAhead[i]); }
    ScoringSystem* sc = car->scoring.get();
    P.p("sc.stepReward", sc->stepReward); P.p("sc.totalReward", sc->totalReward);
    P.p("sc.oldPointId", sc->oldPointId); P.p("sc.oldSplinePointId", sc->oldSplinePointId);
    P.p("sc.drifting", sc->drifting ? 1 : 0); P.p("sc.driftExtreme", sc->driftExtreme ? 1 : 0);
    P.p("sc.driftInvalid", sc->driftInvalid ? 1 : 0); P.p("sc.currentDriftAngle", sc->currentDriftAngle);
    P.p("sc.currentSpeedMultiplier", sc->currentSpeedMultiplier); P.p("sc.lastDriftDirection", sc->lastDriftDirection);
    P.p("sc.driftStraightTimer", sc->driftStraightTimer); P.p("sc.instantDriftDelta", sc->instantDriftDelta);
    P.p("sc.instantDrift", sc->instantDrift); P.p("sc.driftPoints", sc->driftPoints);
    P.p("sc.driftComboCounter", sc->driftComboCounter);
    P.p("car.collisionFlag", car->collisionFlag ? 1 : 0); P.p("car.outOfTrackFlag", car->outOfTrackFlag ? 1 : 0);
    for (int i = 0; i < 5; ++i) { char nm[40]; snprintf(nm, sizeof(nm), "car.damageZoneLevel%d", i); P.p(nm, car->damageZoneLevel[i]); }
    // CarState (what getCarState hands to python)
    const CarState& cs = *car->state;
    P.p("cs.timestamp", cs.timestamp); P.p("cs.engineRPM", cs.engineRPM); P.p("cs.speedMS", cs.speedMS);
    P.p("cs.gear", cs.gear); P.p("cs.gearGrinding", cs.gearGrinding);
    P.p("cs.trackPointId", cs.trackPointId); P.p("cs.lastTrackPointTimestamp", cs.lastTrackPointTimestamp);
    P.p3("cs.bodyEuler", &cs.bodyEuler.x); P.p3("cs.accG", &cs.accG.x); P.p3("cs.velocity", &cs.velocity.x);
    P.p3("cs.localVelocity", &cs.localVelocity.x); P.p3("cs.angularVelocity", &cs.angularVelocity.x);
    P.p3("cs.localAngularVelocity", &cs.localAngularVelocity.x);
    for (int i = 0; i < 4; ++i) { snprintf(nm, sizeof(nm), "cs.hubMatrix%d", i); P.pn(nm, cs.hubMatrix[i].dim1, 16); }
    for (int i = 0; i < 4; ++i) { snprintf(nm, sizeof(nm), "cs.tyreContacts%d", i); P.p3(nm, &cs.tyreContacts[i].x); }
    P.pn("cs.tyreLoad", cs.tyreLoad.data(), 4); P.pn("cs.tyreAngularSpeed", cs.tyreAngularSpeed.data(), 4);
    P.pn("cs.tyreSlipRatio", cs.tyreSlipRatio.data(), 4); P.pn("cs.tyreNdSlip", cs.tyreNdSlip.data(), 4);
    P.pn("cs.probes", cs.probes.data(), 7); P.pn("cs.lookAhead", cs.lookAhead.data(), 5);
    P.p("cs.stepReward", cs.stepReward); P.p("cs.totalReward", cs.totalReward);
}

// the env's 24-slot observation (projectd_env.py:239-273) from the CarState python reads
static void obsOf(const CarState& s, float* o) {
    int k = 0;
    o[k++] = s.localVelocity.x; o[k++] = s.localVelocity.y; o[k++] = s.localVelocity.z;
    o[k++] = s.localAngularVelocity.x; o[k++] = s.localAngularVelocity.y; o[k++] = s.localAngularVelocity.z;
    for (int i = 0; i < 4; ++i) o[k++] = s.tyreNdSlip[i];
    o[k++] = s.bodyVsTrack; o[k++] = s.velocityVsTrack;
    for (int i = 0; i < 5; ++i) o[k++] = s.lookAhead[i];
    for (int i = 0; i < 7; ++i) o[k++] = s.probes[i];
}

struct Env {
    std::shared_ptr<Simulator> sim;
    Car* car = nullptr;
    Car* car2 = nullptr;      // a second car in the same simulator (twoCar scenarios: PyProjectD.cpp:219-237 addCar once more)
    float dist2 = 0.0f;
    double dt = 1.0 / 333.0;  // projectd_env.py:19
    CarControls dcontrols;    // persistent python-side object (projectd_env.py:135)
    CarControls dcontrols2;
    bool smooth = true;       // setCarControls(sim, car, smooth, controls)

    void setupCar(Car* car, const std::string& model, bool autoClutch, bool autoShift, bool autoBlip);
    void init(const std::string& base, const std::string& track, const std::string& model, bool autoClutch = true, bool autoShift = true, bool autoBlip = true, float secondCarAt = -1.0f) {
        // projectd_env.py:118-136, PyProjectD.cpp:111-137
        sim = std::make_shared<Simulator>();
        sim->simulatorId = 0;
        sim->init(strw(base));
        sim->loadTrack(strw(track));
        car = sim->addCar(strw(model));
        car->teleportByMode(TeleportMode::Start);
        setupCar(car, model, autoClutch, autoShift, autoBlip);
        if (secondCarAt >= 0.0f) {
            dist2 = secondCarAt;
            car2 = sim->addCar(strw(model));
            car2->teleportToSpline(dist2);
            setupCar(car2, model, autoClutch, autoShift, autoBlip);
        }
    }
    void setupCarBody(Car* car, const std::string& model, bool autoClutch, bool autoShift, bool autoBlip) {
        car->teleportOnCollision = false; car->teleportOnBadLocation = false; car->teleportMode = 0;
        car->autoClutch->useAutoOnStart = autoClutch; car->autoClutch->useAutoOnChange = autoClutch;   // PyProjectD.cpp:307-317
        car->autoShift->isActive = autoShift; car->autoBlip->isActive = autoBlip;
        const std::pair<const char*, float> tunes[] = {{"FRONT_BIAS", 55.0f}, {"DIFF_POWER", 30.0f}, {"DIFF_COAST", 30.0f},
            {"FINAL_RATIO", 5.0f}, {"PRESSURE_LF", 28.0f}, {"PRESSURE_RF", 28.0f}, {"PRESSURE_LR", 28.0f}, {"PRESSURE_RR", 28.0f}};
        if (model == "ks_toyota_ae86_drift")
            for (auto& t : tunes) car->setup->setTune(t.first, t.second);
        const std::pair<const char*, float> svars[] = {{"SmoothSteerSpeed", 10.0f}, {"MinBonusSpeed", 5.0f}, {"MaxBonusSpeed", 200.0f},
            {"StallRpm", 300.0f}, {"DirectionThreshold", 0.75f}, {"OutOfTrackThreshold", 0.51f}, {"ApproachDistance", 3.5f},
            {"CriticalDistance", 2.0f}, {"TravelBonus", 0.1f}, {"TravelSplineBonus", 0.01f}, {"DriftBonus", 0.0f}, {"SpeedBonus", 0.0f},
            {"ThrottleBonus", 0.0f}, {"EngineRpmBonus", 0.0f}, {"DirectionBonus", 0.0f}, {"DirectionPenalty", 0.0f},
            {"ObstApproachPenalty", 0.0f}, {"CollisionPenalty", 0.0f}, {"OffTrackPenalty", 0.0f}, {"GearGrindPenalty", 0.0f},
            {"StallPenalty", 0.0f}};
        for (auto& v : svars) car->scoring->config->setVar(v.first, v.second);
    }
    void step(float a0, float a1, float b0 = 0.0f, float b1 = 0.0f) {
        // projectd_env.py:157-171, PyProjectD.cpp:160-180,297-305
        dcontrols.steer = a0;
        dcontrols.gas = pdoracle::envGas(a1);
        car->controls = dcontrols;
        car->smoothSteer = smooth;
        if (car2) { dcontrols2.steer = b0; dcontrols2.gas = pdoracle::envGas(b1); car2->controls = dcontrols2; car2->smoothSteer = smooth; }
        sim->step((float)dt, sim->physicsTime, sim->gameTime);
        sim->physicsTime += dt;
        sim->gameTime += dt;
    }
    void stepControls(const pdoracle::Ctl& c) {
        // PyProjectD.cpp:297-305 with every field of the python-side CarControls object written
        dcontrols.steer = c.steer; dcontrols.clutch = c.clutch; dcontrols.brake = c.brake; dcontrols.handBrake = c.handBrake; dcontrols.gas = c.gas;
        dcontrols.requestedGearIndex = (int8_t)c.requestedGearIndex; dcontrols.gearUp = c.gearUp != 0; dcontrols.gearDn = c.gearDn != 0;
        car->controls = dcontrols;
        car->smoothSteer = smooth;
        sim->step((float)dt, sim->physicsTime, sim->gameTime);
        sim->physicsTime += dt;
        sim->gameTime += dt;
    }
    void reset() {
        // projectd_env.py:216-227
        car->teleportByMode(TeleportMode::Start);
        if (car2) car2->teleportToSpline(dist2);
        step(0.0f, 0.0f);
    }
};
void Env::setupCar(Car* c, const std::string& model, bool autoClutch, bool autoShift, bool autoBlip) { setupCarBody(c, model, autoClutch, autoShift, autoBlip); }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <basePath> <track> <outDir> [car] [-v]\n", argv[0]); return 2; }
    const std::string base = argv[1], only = (std::string(argv[2]) == "all") ? std::string() : std::string(argv[2]), outDir = argv[3];
    std::string model = "ks_toyota_ae86_drift";
    for (int i = 4; i < argc; ++i) { if (!strcmp(argv[i], "-v")) ref_enable_log(true); else model = argv[i]; }
    try {
        for (int sid = 0; sid < pdoracle::kNumScenarios; ++sid) {
            const auto& sc = pdoracle::kScenarios[sid];
            INIReader::flushCache();
            Env env;
            const std::string track = sc.track;
            if (!only.empty() && only != track) continue;
            env.init(base, track, sc.car ? std::string(sc.car) : model, sc.autoClutch != 0, sc.autoShift != 0, sc.autoBlip != 0, sc.twoCar ? pdoracle::kTwoCarDist[sc.twoCar - 1] : -1.0f);
            env.smooth = sc.rawSteer == 0;
            ref_set_collide(env.sim->physics.get(), sc.collide != 0);
            if (sc.autoTele) { env.car->teleportOnCollision = sc.autoTele & 1; env.car->teleportOnBadLocation = (sc.autoTele >> 1) & 1; env.car->teleportMode = (sc.autoTele >> 2) & 3; }   // setCarAutoTeleport, PyProjectD.cpp:286-295
            ref_msvc_srand(1);   // every scenario starts like a fresh process
            if (sc.scoringSet) for (int i = 0; i < pdoracle::kNumScoringSetA; ++i) env.car->scoring->config->setVar(pdoracle::kScoringSetA[i].name, pdoracle::kScoringSetA[i].value);   // PyProjectD.cpp:347-355
            if (sc.tuneSet) for (int i = 0; i < pdoracle::kNumTuneSetA; ++i) env.car->setup->setTune(pdoracle::kTuneSetA[i].name, pdoracle::kTuneSetA[i].value);   // PyProjectD.cpp:328-335
            pdoracle::ProbeFile pf, pf2;
            const int nCars = sc.twoCar ? 2 : 1;
            env.reset();
            {
                Probe P; P.names = &pf.names;
                fillProbe(P, env.sim.get(), env.car, 0, nCars);
                pf.add(-1, 0.0f, 0.0f, P);
                if (sc.twoCar) { Probe Q; Q.names = &pf2.names; fillProbe(Q, env.sim.get(), env.car2, 1, nCars); pf2.add(-1, 0.0f, 0.0f, Q); }
            }
            for (int t = 0; t < sc.ticks; ++t) {
                if (sc.resetEvery && t > 0 && t % sc.resetEvery == 0) {
                    const int k = t / sc.resetEvery - 1;
                    if (sc.teleDist == 1) { env.car->teleportToSpline(pdoracle::kTeleDist[k % 4]); env.step(0.0f, 0.0f); }   // teleportCarToSpline + a zero-action tick
                    else if (sc.teleDist == 2) { env.car->teleportToPits(pdoracle::kTelePit[k % 5]); env.step(0.0f, 0.0f); }   // teleportCarToPits (PyProjectD.cpp:259-266)
                    else if (sc.teleDist == 3) {   // teleportCarToLocation (PyProjectD.cpp:250-257): forcePosition(vec3f(x, y, z))
                        const vec3f p = env.car->body->getPosition(0.0f);
                        const float* o = pdoracle::kTeleLoc[k % 4];
                        env.car->forcePosition(vec3f(p.x + o[0], p.y + o[1], p.z + o[2]));
                        env.step(0.0f, 0.0f);
                    }
                    else env.reset();
                }
                if (sc.boostAt && t == sc.boostAt) {   // through the engine seam: every body, as the oracle edits the state record
                    pdrb::World* w = ref_get_world(env.sim->physics.get());
                    for (auto& b : w->bodies) b.lvel[2] = 50.0f;
                }
                float a0, a1, b0 = 0.0f, b1 = 0.0f;
                if (sc.feedback) {
                    float obs[24]; obsOf(*env.car->state, obs);
                    pdoracle::scenarioFeedback(sid, t, obs, a0, a1);
                    if (sc.twoCar) { float obs2[24]; obsOf(*env.car2->state, obs2); pdoracle::scenarioFeedback2(sid, t, obs2, b0, b1); }
                    env.step(a0, a1, b0, b1);
                } else if (sc.twoCar) {
                    pdoracle::scenarioAction(sid, t, a0, a1);
                    pdoracle::scenarioAction2(sid, t, b0, b1);
                    env.step(a0, a1, b0, b1);
                } else if (sc.full) {
                    pdoracle::Ctl c; pdoracle::scenarioControls(sid, t, c);
                    a0 = c.steer; a1 = c.gas;
                    env.stepControls(c);
                } else {
                    pdoracle::scenarioAction(sid, t, a0, a1);
                    env.step(a0, a1);
                }
                if (pdoracle::scenarioRecord(sc, t)) {
                    Probe P;
                    fillProbe(P, env.sim.get(), env.car, 0, nCars);
                    pf.add(t, a0, a1, P);
                    if (sc.twoCar) { Probe Q; fillProbe(Q, env.sim.get(), env.car2, 1, nCars); pf2.add(t, b0, b1, Q); }
                }
            }
            const std::string path = outDir + "/" + track + "_" + sc.name + ".bin";
            if (!pf.write(path.c_str())) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 1; }
            fprintf(stderr, "wrote %s: %d records x %d fields\n", path.c_str(), (int)pf.ticks.size(), pf.nfields);
            if (sc.twoCar) {
                const std::string path2 = outDir + "/" + track + "_" + sc.name + "_b.bin";
                if (!pf2.write(path2.c_str())) { fprintf(stderr, "cannot write %s\n", path2.c_str()); return 1; }
            }
        }
    } catch (const std::exception& ex) {
        fprintf(stderr, "EXCEPTION: %s\n", ex.what());
        return 1;
    }
    return 0;
}
