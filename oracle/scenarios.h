// ORACLE / TEST INFRASTRUCTURE.  Scripted action sequences shared by the reference-TU harness
// (oracle/refharness) and the CPU restatement (oracle/cpu_ref).  Actions are the env's 2-vector
// (pyprojectd/projectd_env.py:157-160): a0 = steer in [-1,1], a1 -> gas = linscale(a1,-1,1,0.1,1).
#pragma once
#include <cmath>

namespace pdoracle {

// full: 0 = env 2-vector (steer, a1) with the env's assists (all on); 1 = every CarControls field scripted
// (PyProjectD.cpp:297-305 setCarControls) with the scenario's own assist switches (setCarAssists :307-317)
// track: synthetic track the scenario runs on (projectd-core_amd/synthetic_tracks.py); feedback: 1 = the action is a function of the
// previous tick's 24-slot observation (scenarioFeedback), i.e. a closed loop like an RL policy
// car: model directory under content/cars (nullptr = the env default, ks_toyota_ae86_drift); rawSteer: 1 = setCarControls(smooth = false)
struct Scenario { const char* name; int ticks; int denseTicks; int stride; int full; int autoClutch, autoShift, autoBlip; const char* track; int feedback; const char* car; int rawSteer;
                  int collide; /* run the engine's collision pass (the other fixtures predate it and keep it off) */
                  int resetEvery; /* env.reset() (teleportByMode(Start) + one zero-action tick, projectd_env.py:216-227) every so many ticks */
                  int tuneSet; /* apply kTuneSetA through setCarTune after the env's own tunes */
                  int teleDist; /* 1: the resets teleport to kTeleDist[k % 4] along the spline (teleportCarToSpline) instead of to the start; 2: to pit box kTelePit[k % 5]
                                   (teleportCarToPits); 3: to the chassis' position + kTeleLoc[k % 4] (teleportCarToLocation = Car::forcePosition) */
                  int scoringSet; /* kScoringSetA through setScoringVar: every reward weight and threshold non-default and non-zero */
                  int boostAt; /* before this tick every body's linear velocity z is set to 50 m/s (180 km/h); 0 = never */
                  int autoTele; /* setCarAutoTeleport: bit 0 on collision, bit 1 on bad location, bits 2-3 mode (0 Start, 1 Nearest, 2 Random) */
                  int twoCar; /* a second car of the same model in the same simulator (Simulator::addCar twice), put down kTwoCarDist[twoCar - 1] along the spline ahead of the first;
                                 its script: scenarioAction2 / scenarioFeedback2; its records go to a second probe file (<track>_<name>_b) */ };

static const Scenario kScenarios[] = {
    {"idle", 600, 200, 10, 0, 1, 1, 1, "flat", 0, nullptr, 0, 0, 0, 0, 0, 0, 0},
    {"launch", 2000, 450, 10, 0, 1, 1, 1, "flat", 0, nullptr, 0, 0, 0, 0, 0, 0, 0},
    {"circle", 1600, 300, 10, 0, 1, 1, 1, "flat", 0, nullptr, 0, 0, 0, 0, 0, 0, 0},
    {"slalom", 2400, 300, 10, 0, 1, 1, 1, "flat", 0, nullptr, 0, 0, 0, 0, 0, 0, 0},
    {"brake", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, nullptr, 0, 0, 0, 0, 0, 0, 0},    // pedal brake to a stop, handbrake turn (BrakeSystem, tyre lock)
    {"manual", 2600, 300, 10, 1, 0, 0, 0, "flat", 0, nullptr, 1, 0, 0, 0, 0, 0, 0},   // no assists: manual clutch, gearUp/gearDn pulses, H-shifter gear select, grinding
    {"drive", 7000, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 0, 0, 0, 0, 0},   // closed hilly, banked mountain road driven by a probe-feedback controller (configs[2] shape)
    {"rx7", 2400, 300, 10, 0, 1, 1, 1, "flat", 0, "ks_mazda_rx7_tuned", 0, 0, 0, 0, 0, 0, 0},          // double wishbones all round, one turbo: the slalom script
    {"supra", 5000, 300, 10, 0, 1, 1, 1, "touge", 1, "ks_toyota_supra_mkiv_drift", 0, 0, 0, 0, 0, 0, 0},   // double wishbones, two turbos, 6 gears, on the mountain road
    {"fc3s", 2400, 300, 10, 0, 1, 1, 1, "flat", 0, "dthwsh_mazda_rx7_fc3s_sr20", 0, 0, 0, 0, 0, 0, 0},      // strut front + double wishbone rear (38 rows), 3 wings + 2 fins, turbo
    {"readie", 5000, 300, 10, 0, 1, 1, 1, "touge", 1, "gravygarage_street_ae86_readie", 0, 0, 0, 0, 0, 0, 0},   // strut front + double wishbone rear on the mountain road
    {"playground", 2500, 300, 10, 0, 1, 1, 1, "driftplayground", 1, nullptr, 0, 0, 0, 0, 0, 0, 0},   // the env's default track as shipped (510 surfaces, 112 411 triangles, spline.cache)
    {"multilink", 4000, 300, 10, 0, 1, 1, 1, "touge", 1, "pdb_ml_supra", 0, 0, 0, 0, 0, 0, 0},   // reference SuspensionML on a derived car (oracle/make_base.py), front and rear
    {"heave", 4000, 300, 10, 0, 1, 1, 1, "touge", 1, "pdb_heave_rx7", 0, 0, 0, 0, 0, 0, 0},       // reference HeaveSpring on a derived car (third spring across both axles)
    {"fwd", 2400, 300, 10, 0, 1, 1, 1, "flat", 0, "pdb_fwd_ae86", 0, 0, 0, 0, 0, 0, 0},            // front-wheel drive through the same 2WD drivetrain (derived car), slalom script
    // body contacts: full throttle down the walled strip -- the belly box scrapes the ridge, then the car drifts into the side wall
    // (hull).  Pins what the reference does with a contact (Simulator / Car::onCollisionCallback, the scoring that reads the flag
    // and the damage); the contacts themselves are this project's (oracle/rb/pdcollide.h).  Stride 3: odd and even frames alternate.
    {"walled", 2150, 0, 3, 0, 1, 1, 1, "walled", 0, nullptr, 0, 1, 0, 0, 0, 0, 0},
    // episode resets in mid-flight (Car::teleportByMode(Start) -> teleportToSpline -> forceRotation / forcePosition -> Car::reset,
    // Tyre::reset, Drivetrain::reset, suspension attach: Car.cpp:385-410,1240-1358): the car is driven on the mountain road and
    // reset every 700 ticks from whatever state it is in (rolling, warm tyres, a gear engaged, turbos spun up)
    {"resets", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 700, 0, 0, 0, 0},
    {"resets_supra", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, "ks_toyota_supra_mkiv_drift", 0, 0, 700, 0, 0, 0, 0},
    {"resets_fc3s", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, "dthwsh_mazda_rx7_fc3s_sr20", 0, 0, 650, 0, 0, 0, 0},
    // SetupManager (Car/SetupManager.cpp:10-330): a broad set of setCarTune calls -- in and out of range, on and off the step
    // grid, names a car's setup.ini does not list -- on a strut / live-axle car and on a strut / double-wishbone car with wings
    {"tunes", 2400, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 0, 1, 0, 0, 0},
    {"tunes_fc3s", 2400, 300, 10, 0, 1, 1, 1, "touge", 1, "dthwsh_mazda_rx7_fc3s_sr20", 0, 0, 0, 1, 0, 0, 0},
    {"tunes_supra", 2400, 300, 10, 0, 1, 1, 1, "touge", 1, "ks_toyota_supra_mkiv_drift", 0, 0, 0, 1, 0, 0, 0},   // two adjustable turbos: TURBO_n
    // Car::teleportToSpline at arbitrary distances, in mid-flight (teleportCarToSpline, PyProjectD.cpp:274-281)
    {"teleports", 2600, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 500, 0, 1, 0, 0},
    // ScoringSystem::computeAgentReward with every weight in play (the env zeroes most of them): on the road and then off it,
    // through the manual-gearbox script (grinding, stalling), and down the walled strip (collision penalty)
    {"rewards", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 0, 0, 0, 1, 0},
    {"rewards_manual", 2600, 300, 10, 1, 0, 0, 0, "flat", 0, nullptr, 1, 0, 0, 0, 0, 1, 0},
    {"rewards_walled", 2150, 0, 3, 0, 1, 1, 1, "walled", 0, nullptr, 0, 1, 0, 0, 0, 1, 0},
    // the other shipped tracks that come with their mesh (build container only): real surface kinds, sectors, pit lanes, traced sides
    {"ebisu", 3000, 300, 10, 0, 1, 1, 1, "ebisu_touge", 1, nullptr, 0, 0, 0, 0, 0, 0, 0},
    {"yamanashi", 3000, 300, 10, 0, 1, 1, 1, "yamanashi_short", 1, "ks_mazda_rx7_tuned", 0, 0, 0, 0, 0, 0, 0},
    {"euphoria", 3000, 300, 10, 0, 1, 1, 1, "euphoria_hillside_park", 1, "gravygarage_street_ae86_readie", 0, 0, 0, 0, 0, 0, 0},
    // into the wall across the road at 180 km/h: Engine::blowUp above 150 (Car.cpp:979-980) and the dead engine afterwards
    {"crash", 2100, 0, 3, 0, 1, 1, 1, "walled", 0, nullptr, 0, 1, 0, 0, 0, 0, 1650},
    // collision RESPONSE (contact joints, oracle/rb/pdrb.cpp solveContacts): a gentle lock takes the car into the side wall and along
    // it (hull contacts with friction, many ticks in contact); then full throttle into the wall across the road, again and again
    {"scrape", 3000, 0, 3, 0, 1, 1, 1, "walled", 0, nullptr, 0, 1, 0, 0, 0, 0, 0},
    {"wallpush", 3900, 0, 3, 0, 1, 1, 1, "walled", 0, "ks_toyota_supra_mkiv_drift", 0, 1, 0, 0, 0, 0, 0},
    // setCarAutoTeleport (PyProjectD.cpp:286-295): Car::teleportByMode INSIDE the tick (ScoringSystem.cpp:194-225) -- on the
    // walled strip back to the nearest spline point at every wall contact; on the mountain road to a random point of the lap
    // (the C runtime's rand()) whenever the fixed lock takes the car off the road; full throttle into the wall, back to the start
    {"autotele_near", 3000, 0, 3, 0, 1, 1, 1, "walled", 0, nullptr, 0, 1, 0, 0, 0, 0, 0, 1 | 2 | (1 << 2)},
    {"autotele_rand", 3600, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 0, 0, 0, 0, 0, 2 | (2 << 2)},
    {"autotele_start", 3900, 0, 3, 0, 1, 1, 1, "walled", 0, nullptr, 0, 1, 0, 0, 0, 0, 0, 1 | (0 << 2)},
    // the two tracks of BASELINE configs[2] / [4], which ship their spline but not their mesh (build container only): the reference's
    // own spline.bin -- 5109 points at 0.9 m, 13 323 points at 1.6 m, open, racing-line sides -- with a road ribbon generated around
    // it (synthetic_tracks.gen_ribbon); the second Akina run is put down at four places along the hill (teleportCarToSpline)
    {"akina", 3000, 300, 10, 0, 1, 1, 1, "ek_akina", 1, nullptr, 0, 0, 0, 0, 0, 0, 0},
    {"akina_tele", 3000, 300, 10, 0, 1, 1, 1, "ek_akina", 1, "gravygarage_street_ae86_readie", 0, 0, 600, 0, 1, 0, 0},
    {"nords", 3000, 300, 10, 0, 1, 1, 1, "ks_nordschleife", 1, "ks_toyota_supra_mkiv_drift", 0, 0, 0, 0, 0, 0, 0},
    // branches no shipped car takes, on a derived car (oracle/make_base.py): [THROTTLE_RESPONSE], [COAST_SETTINGS], [EBB] -- the brake
    // script (full and part throttle, a hard stop from speed, a hand-brake turn, trail braking)
    {"cold", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_cold_rx7", 0, 0, 0, 0, 0, 0, 0},
    // the tyre's LUT forms (DY_CURVE / DX_CURVE / DCAMBER_LUT through the cubic spline of Curve::getCubicSplineValue) on a derived car, driven
    // round the mountain road: loads from nothing to twice the static one, camber of both signs
    {"curves", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, "pdb_curves_ae86", 0, 0, 0, 0, 0, 0, 0},
    // wings with ground-effect LUTs (LUT_GH_CL / LUT_GH_CD over Car::getPointGroundHeight) on a derived car, round the mountain road
    {"groundfx", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, "pdb_gh_fc3s", 0, 0, 0, 0, 0, 0, 0},
    // an aero.ini without wings: AeroMap's own drag and lift from [DATA] (derived car), round the mountain road
    {"aerodata", 3000, 300, 10, 0, 1, 1, 1, "touge", 1, "pdb_aerodata_ae86", 0, 0, 0, 0, 0, 0, 0},
    // wing dynamic controllers ([DYNAMIC_CONTROLLER_n]: speed, brake, lateral g and throttle moving three wings' angles) on a derived car -- the brake script
    {"wingctrl", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_wingctrl_fc3s", 0, 0, 0, 0, 0, 0, 0},
    // DynamicController files on a derived car (a turbo's wastegate, the other's maxBoost, the differential's preload) -- the brake script
    {"dynctrl", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_dynctrl_supra", 0, 0, 0, 0, 0, 0, 0},
    // the brake system's controller files (ctrl_ebb.ini: the front bias; steer_brake_controller.ini: torque on the inner rear wheel) -- the brake script
    {"brakectrl", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_brakectrl_rx7", 0, 0, 0, 0, 0, 0, 0},
    // the controller inputs that read the tyres' status (slip ratios and angles, oversteer factor, wheel-speed ratio, load spread, steering angles), on two derived cars
    {"ctrlin_a", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_ctrlin_a_ae86", 0, 0, 0, 0, 0, 0, 0},
    {"ctrlin_b", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_ctrlin_b_ae86", 0, 0, 0, 0, 0, 0, 0},
    // brake disc temperatures ([TEMPS_FRONT] / [TEMPS_REAR]) on a derived car: the brake script, with a reset to the start in mid-run (the discs go to the ambient temperature)
    {"braketemp", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_braketemp_rx7", 0, 0, 1300, 0, 0, 0, 0},
    // the wing controllers' other inputs (rear suspension travel, longitudinal g, steer) on a derived car -- the brake script
    {"wingctrl2", 2400, 200, 10, 1, 1, 1, 1, "flat", 0, "pdb_wingctrl2_fc3s", 0, 0, 0, 0, 0, 0, 0},
    // a second box collider (CarColliderManager.cpp:17-33 takes every COLLIDER_n of colliders.ini; every shipped car has one): the AE86 with a front
    // splitter box that hangs lower than the belly box, full throttle down the walled strip -- the splitter meets the ridge first, then both boxes scrape
    {"twobox", 2150, 0, 3, 0, 1, 1, 1, "walled", 0, "pdb_twobox_ae86", 0, 1, 0, 0, 0, 0, 0},
    // teleportCarToPits (PyProjectD.cpp:259-266 -> Car::teleportToPits -> Car::teleport(pit matrix): Car.cpp:1310-1323, the boxes of pits.ini: Track.cpp:151-175) in
    // mid-flight: on the mountain road (pits.ini written by synthetic_tracks.gen_touge: boxes on the road, beside it, one with nothing under it, headings of
    // both signs and beyond a turn) and on the env's default track with the pits.ini it ships; one id of the schedule is outside the list (the reference does nothing)
    {"pits", 3600, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 700, 0, 2, 0, 0},
    {"pits_playground", 3600, 300, 10, 0, 1, 1, 1, "driftplayground", 1, nullptr, 0, 0, 700, 0, 2, 0, 0},
    // teleportCarToLocation (PyProjectD.cpp:250-257 -> Car::forcePosition, Car.cpp:1240-1272) in mid-flight: the car keeps whatever orientation it has, is put down
    // on what the ray from 10 m above the point meets (or left at the point's height when it meets nothing: the last offset leaves the mountain road)
    {"locations", 3600, 300, 10, 0, 1, 1, 1, "touge", 1, nullptr, 0, 0, 800, 0, 3, 0, 0},
    {"locations_fc3s", 2400, 300, 10, 0, 1, 1, 1, "flat", 0, "dthwsh_mazda_rx7_fc3s_sr20", 0, 0, 550, 0, 3, 0, 0},
    // two cars in one simulator (cfg/sim.ini ships MAX_CARS = 2): Car::updateAirPressure (Car.cpp:557-585) thins the air a car meets by the wakes of the OTHER cars
    // (Sim/SlipStream.cpp), as their last Car::postStep left them.  On a derived car whose aero.ini carries [SLIPSTREAM] (EFFECT_GAIN_MULT 1.5, SPEED_FACTOR_MULT 4: no
    // shipped car has the section): on the plane the car behind, flat out and weaving a little, closes on the slower car ahead through its wake, drives through it (the
    // harness's engine collides cars with the track only) and tows it in turn; the same with two stock AE86 (SlipStream.h's defaults: a wake of a quarter of a second's travel)
    {"twocar_draft", 4200, 300, 10, 0, 1, 1, 1, "flat", 0, "pdb_slip_ae86", 0, 0, 0, 0, 0, 0, 0, 0, 1},
    {"twocar_stock", 4200, 300, 10, 0, 1, 1, 1, "flat", 0, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 1},
};
static const int kNumScenarios = 56;
static const float kTwoCarDist[2] = {0.0034f, 0.0035f};
static const float kTeleDist[4] = {0.13f, 0.41f, 0.77f, 0.95f};
static const int kTelePit[5] = {0, 2, 99, 1, 3};
static const float kTeleLoc[4][3] = {{0.7f, 5.0f, -0.4f}, {-1.1f, 0.3f, 0.9f}, {0.0f, 12.0f, 0.0f}, {30.0f, 2.0f, 30.0f}};
struct ScoreVar { const char* name; float value; };
static const ScoreVar kScoringSetA[] = {
    {"SmoothSteerSpeed", 7.0f}, {"MinBonusSpeed", 8.0f}, {"MaxBonusSpeed", 150.0f}, {"StallRpm", 900.0f}, {"DirectionThreshold", 0.6f},
    {"OutOfTrackThreshold", 0.45f}, {"ApproachDistance", 6.0f}, {"CriticalDistance", 1.5f}, {"TravelBonus", 0.3f}, {"TravelSplineBonus", 0.02f},
    {"DriftBonus", 0.07f}, {"SpeedBonus", 0.5f}, {"ThrottleBonus", 0.11f}, {"EngineRpmBonus", 0.13f}, {"DirectionBonus", 0.21f},
    {"DirectionPenalty", 0.17f}, {"ObstApproachPenalty", 0.9f}, {"CollisionPenalty", 3.0f}, {"OffTrackPenalty", 2.0f}, {"GearGrindPenalty", 0.7f},
    {"StallPenalty", 0.4f},
};
static const int kNumScoringSetA = (int)(sizeof(kScoringSetA) / sizeof(kScoringSetA[0]));
struct Tune { const char* name; float value; };
static const Tune kTuneSetA[] = {
    {"ARB_FRONT", 22000.0f}, {"ARB_REAR", 5000.0f}, {"BRAKE_POWER_MULT", 93.0f}, {"CAMBER_LF", -2.0f}, {"CAMBER_RF", -3.5f}, {"CAMBER_LR", -1.5f},
    {"CAMBER_RR", -1.0f}, {"DAMP_BUMP_LF", 5100.0f}, {"DAMP_BUMP_RF", 9000.0f}, {"DAMP_BUMP_LR", 6000.0f}, {"DAMP_BUMP_RR", 6500.0f},
    {"DAMP_FAST_BUMP_LF", 8000.0f}, {"DAMP_FAST_BUMP_RR", 3000.0f}, {"DAMP_REBOUND_LF", 9300.0f}, {"DAMP_REBOUND_RR", 8800.0f},
    {"DAMP_FAST_REBOUND_LF", 13000.0f}, {"DAMP_FAST_REBOUND_LR", 9000.0f}, {"DIFF_PRELOAD", 55.0f}, {"DIFF_POWER", 65.0f}, {"DIFF_COAST", 45.0f},
    {"FRONT_BIAS", 61.0f}, {"PRESSURE_LF", 24.0f}, {"PRESSURE_RR", 33.0f}, {"ROD_LENGTH_LF", -120.0f}, {"ROD_LENGTH_RF", -95.0f},
    {"ROD_LENGTH_LR", -330.0f}, {"ROD_LENGTH_RR", -310.0f}, {"SPRING_RATE_LF", 82.0f}, {"SPRING_RATE_RR", 71.0f}, {"TOE_OUT_LF", 20.0f},
    {"TOE_OUT_RF", -35.0f}, {"TOE_OUT_LR", 15.0f}, {"FINAL_RATIO", 4.5f}, {"ENGINE_LIMITER", 95.0f}, {"WING_0", 5.0f}, {"WING_3", 12.0f},
    {"INTERNAL_GEAR_2", 2.0f}, {"BUMP_STOP_RATE_LF", 80.0f}, {"PACKER_RANGE_LF", 60.0f}, {"PROGRESSIVE_SPRING_RATE_LF", 10.0f}, {"FUEL", 20.0f},
    {"TURBO_0", 0.35f}, {"TURBO_1", 0.8f}, {"TURBO_2", 0.5f}, {"NO_SUCH_TUNE", 1.0f},
};
static const int kNumTuneSetA = (int)(sizeof(kTuneSetA) / sizeof(kTuneSetA[0]));
#define PDORACLE_DEFAULT_CAR "ks_toyota_ae86_drift"

// closed-loop action from the previous observation (projectd_env.py:239-273 slot order): centre between the side probes,
// align with the +-25 degree probes, damp with the yaw rate, hold ~12 m/s.  Plain float arithmetic, fixed order.
inline void scenarioFeedback(int sid, int tick, const float* obs, float& a0, float& a1) {
    const float lat = obs[21] - obs[20];      // probes[4] - probes[3]  (-90 / +90 degrees, 10 m)
    const float head = obs[19] - obs[18];     // probes[2] - probes[1]  (-25 / +25 degrees, 50 m)
    const float yaw = obs[4];                 // localAngularVelocity.y
    const float v = obs[2];                   // localVelocity.z
    float s = (0.03f * lat + 0.015f * head) + 0.15f * yaw;
    if (s < -1.0f) s = -1.0f;
    if (s > 1.0f) s = 1.0f;
    float g = 0.3f * (12.0f - v);
    if (g < -1.0f) g = -1.0f;
    if (g > 1.0f) g = 1.0f;
    if (sid == 33 && (tick % 900) > 600) s = 0.45f;   // `autotele_rand`: every so often a fixed lock takes the car off the road
    if (sid == 23 && tick > 1500) s = 0.45f;   // `rewards`: then a fixed lock takes the car off the road, to meet the off-track / direction terms
    a0 = s; a1 = g;
}

// the second car's scripts (twoCar scenarios)
inline void scenarioAction2(int sid, int tick, float& a0, float& a1) { (void)sid; (void)tick; a0 = 0.0f; a1 = -0.5f; }   // straight on, a third of the throttle
inline void scenarioFeedback2(int sid, int tick, const float* obs, float& a0, float& a1) {
    scenarioFeedback(sid, tick, obs, a0, a1);
    float g = 0.3f * (9.0f - obs[2]);   // three m/s slower than the first car's law: the car behind closes in
    if (g < -1.0f) g = -1.0f;
    if (g > 1.0f) g = 1.0f;
    a1 = g;
}

struct Ctl { float steer, clutch, brake, handBrake, gas; int requestedGearIndex, gearUp, gearDn; };

inline void scenarioAction(int sid, int tick, float& a0, float& a1) {
    const double t = (double)tick * (1.0 / 333.0);
    switch (sid) {
    case 0: a0 = 0.0f; a1 = -1.0f; break;
    case 1: case 15: case 25: case 29: case 49: a0 = 0.0f; a1 = 1.0f; break;
    case 2: a0 = 0.35f; a1 = 0.2f; break;
    case 30: case 32: a0 = 0.04f; a1 = 0.8f; break;
    case 31: case 34: a0 = 0.0f; a1 = 1.0f; break;
    case 54: case 55: a0 = (float)(0.012 * cos(6.283185307179586 * t / 3.0)); a1 = 1.0f; break;   // `twocar_*`, the car behind: flat out, weaving a little about the line
    default:   // slalom (3), the rx7 run (7), the fc3s run (9)
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
    if (sid < 4 || sid == 7 || sid == 9 || sid == 14) { float a0, a1; scenarioAction(sid, tick, a0, a1); c.steer = a0; c.gas = envGas(a1); return; }
    if (sid == 4 || sid == 38 || (sid >= 42 && sid <= 48)) {
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
