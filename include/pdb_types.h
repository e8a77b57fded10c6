/* pdbatch -- MI355X-native batched vehicle-dynamics stepper (ProjectD-compatible).
 *
 * Plain-C data formats shared by the host library, the HIP kernels and (as documented input /
 * output data) the test oracle.  All structs are POD with fixed layout; no pointers.
 *
 * Reference anchors (relative to /root/reference/src/ProjectD):
 *   pdb_controls   <- Car/CarControls.h:9-20      (24 B, #pragma pack(4))
 *   pdb_car_state  <- Car/CarState.h:11-56        (664 B, what getCarState copies to Python)
 *   pdb_car_params <- every "config" block of the Car/ headers that Car::step reads
 *   pdb_dyn_state  <- every `// runtime` member that survives from one tick to the next
 */
#ifndef PDB_TYPES_H
#define PDB_TYPES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDB_MAX_CURVE 24
#define PDB_MAX_BODIES 8
#define PDB_MAX_JOINTS 24
#define PDB_MAX_ROWS 40   /* 33: strut front / live axle rear; 26: double wishbones; 38: strut front / double wishbone rear */
#define PDB_MAX_WINGS 6
#define PDB_MAX_GEARS 10
#define PDB_NUM_PROBES 7
#define PDB_NUM_LOOKAHEAD 5
#define PDB_OBS_DIM 24

/* piecewise-linear LUT with clamped ends (Core/Curve.cpp:94-115) */
typedef struct pdb_curve {
    int32_t n;
    float x[PDB_MAX_CURVE];
    float y[PDB_MAX_CURVE];
} pdb_curve;

#define PDB_MAX_WING_CTRL 4
#define PDB_MAX_CTRL_STAGES 8
/* natural cubic spline through a LUT's points (Core/Curve.cpp:117-126 getCubicSplineValue -> Core/CubicSpline.cpp -> the tk::spline header the
 * reference vendors, float arithmetic): f(x) = ((a_i h + b_i) h + c_i) h + y_i, h = x - x_i, i = the last point below x; quadratic continuation
 * outside the points (left: b0, c0; right: b_{n-1}, c_{n-1}).  The coefficients are worked out once, by the loader.  n = 0: no curve. */
#define PDB_MAX_SPLINE 16
typedef struct pdb_spline {
    int32_t n;
    float b0, c0;
    int32_t pad;
    float x[PDB_MAX_SPLINE], y[PDB_MAX_SPLINE], a[PDB_MAX_SPLINE], b[PDB_MAX_SPLINE], c[PDB_MAX_SPLINE];
} pdb_spline;

/* Car/CarControls.h:9-20 */
#pragma pack(push, 4)
typedef struct pdb_controls {
    float steer, clutch, brake, handBrake, gas;
    int8_t isShifterSupported, requestedGearIndex, gearUp, gearDn;
} pdb_controls;

/* Car/CarState.h:11-56 -- identical field order and packing */
typedef struct pdb_car_state {
    int32_t carId, simId;
    float timestamp;
    pdb_controls controls;
    int32_t collisionFlag, outOfTrackFlag, trackPointId;
    float lastTrackPointTimestamp, trackLocation, bodyVsTrack, velocityVsTrack;
    float engineRPM, speedMS;
    int32_t gear, gearGrinding;
    float bodyMatrix[16];
    float bodyPos[3], bodyEuler[3], accG[3], velocity[3], localVelocity[3], angularVelocity[3], localAngularVelocity[3];
    float hubMatrix[4][16];
    float tyreContacts[4][3];
    float tyreLoad[4], tyreAngularSpeed[4], tyreSlipRatio[4], tyreNdSlip[4];
    float probes[10];
    float lookAhead[5];
    float stepReward, totalReward;
} pdb_car_state;
#pragma pack(pop)

/* ---------------------------------------------------------------------------------------------
 * Per-model constant block
 * ------------------------------------------------------------------------------------------- */
enum { PDB_JOINT_FIXED = 0, PDB_JOINT_BALL = 1, PDB_JOINT_SLIDER = 2, PDB_JOINT_DBALL = 3 };
enum { PDB_SUSP_DW = 0, PDB_SUSP_STRUT = 1, PDB_SUSP_AXLE = 2, PDB_SUSP_ML = 3 };
/* body slots */
enum { PDB_BODY_CHASSIS = 0, PDB_BODY_TANK = 1, PDB_BODY_AXLE = 2, PDB_BODY_HUB0 = 3, PDB_BODY_STRUT0 = 4,
       PDB_BODY_HUB1 = 5, PDB_BODY_STRUT1 = 6 };

typedef struct pdb_body_def {
    float mass;
    float inertia[3];   /* diagonal of the body-frame inertia (dMassSetBoxTotal) */
} pdb_body_def;

/* joints are stored in SOLVER ROW ORDER (ODE island traversal order) */
typedef struct pdb_joint_def {
    int32_t type, b0, b1;
    int32_t suspErp;     /* 1: Car::step rewrites ERP/CFM each tick (Car.cpp:426-448) */
    int32_t steerWheel;  /* 0/1: this is wheel i's tie-rod DBall re-seated each tick; else -1 */
    float erp, cfm;
    float anchor1[3], anchor2[3];
    float axis1[3], offset[3], qrel[4];
    float distance;      /* DBall stored distance (JointODE.cpp:59) */
} pdb_joint_def;

typedef struct pdb_damper {
    float bumpSlow, reboundSlow, bumpFast, reboundFast, fastThresholdBump, fastThresholdRebound;
} pdb_damper;

typedef struct pdb_susp {
    int32_t type;
    int32_t hubBody, strutBody;          /* body slots (axle: hubBody = axle) */
    float k, progressiveK, bumpStopUp, bumpStopDn, bumpStopRate, rodLength, toeOutLinear, staticCamber, packerRange;
    pdb_damper damper;
    float basePosition[3];
    /* strut (Car/SuspensionStrut.h:9-21,53-67) */
    float carStrut[3];                   /* dataRelToBody.carStrut   (body-local) */
    float tyreStrut[3];                  /* dataRelToWheel.tyreStrut (hub-local)  */
    float tyreSteer[3];                  /* dataRelToWheel.tyreSteer (hub-local)  */
    float baseCarSteer[3];               /* baseCarSteerPosition     (body-local) */
    float refPointY, refPointSignX;
    float strutBaseLength, strutBodyLength;
    /* axle (Car/SuspensionAxle.h:52-62) */
    float axleTrack, referenceY, attachRelativePos, leafSpringKx;
    float axleBasePos[3];
    float sideSign;                      /* +1 left, -1 right */
    float mass;                          /* ISuspension::getMass() */
    /* double wishbone (Car/SuspensionDW.h): spring / damper / bump stops act between basePosition and the hub along the body's up axis */
    float bumpStopProgressive;
} pdb_susp;

typedef struct pdb_heave {               /* Car/HeaveSpring.h: third spring/damper across a double-wishbone axle; active when k != 0 */
    float k, progressiveK, bumpStopUp, bumpStopDn, rodLength, bumpStopRate, packerRange;
    pdb_damper damper;
    int32_t _pad;
} pdb_heave;

#define PDB_MAX_TURBOS 3
typedef struct pdb_turbo {               /* Car/Turbo.h TurboDef + Turbo::userSetting */
    float lagDN, lagUP, maxBoost, wastegate, rpmRef, gamma, userSetting;
    int32_t isAdjustable;
} pdb_turbo;

typedef struct pdb_tyre {
    /* TyreData / TyreModelData of compound 0 (Car/TyreCompound.h:9-75), after Tyre::setCompound */
    float radius, rimRadius, k, d, angularInertia, thermalFrictionK, thermalRollingK, thermalRollingSurfaceK;
    float radiusRaiseK, softnessIndex;
    float Fz0;        /* SCTM::Fz0 = [FRONT|REAR] FZ0 (Tyre.cpp:369) */
    float modelFz0;   /* TyreModelData::Fz0: never assigned by Tyre::initCompounds, stays at its default 2000 (TyreCompound.h:37); used by the relaxation-length filter (TyreForces.cpp:49) */
    float relaxationLength, rr0, rr1, rr_slip, pressureRef, pressureSpringGain, pressureRRGain, pressureGainD, idealPressure;
    float flatSpotK, explosionTemperature, pressureTemperatureGain, pressureStatic;
    int32_t version, driven;
    /* SCTM (Car/TyreModel.h:18-39) */
    float lsMultY, lsExpY, lsMultX, lsExpX, maxSlip0, maxSlip1, asy, falloffSpeed, speedSensitivity, camberGain;
    float dcamber0, dcamber1, cfXmult, pressureCfGain, brakeDXMod, dCamberBlend, combinedFactor;
    /* thermal (Car/TyreThermalModel.h, Car/TyreCompound.h:77-84) */
    float surfaceTransfer, patchTransfer, patchCoreTransfer, internalCoreTransfer, coolFactorGain, camberSpreadK;
    pdb_curve performanceCurve;
    pdb_curve wearCurve;
    /* load-sensitivity and camber LUTs that replace the closed forms when a compound carries them (Tyre.cpp:161-165,207-211; TyreModel.cpp:49-57,121-146):
     * curveFlags bit 0 = DY_CURVE, bit 1 = DX_CURVE (both through the cubic spline), bit 2 = DCAMBER_LUT, bit 3 = DCAMBER_LUT_SMOOTH (spline instead
     * of the piecewise-linear reading; the spline's x / y hold the LUT either way) */
    int32_t curveFlags, curvePad;
    pdb_spline dyLoadCurve, dxLoadCurve, dCamberCurve;
} pdb_tyre;

typedef struct pdb_wing {
    float position[3];
    float area, cdGain, clGain, yawGain, angle;
    int32_t isVertical;
    pdb_curve lutAOA_CL, lutAOA_CD;
    pdb_curve lutGH_CL, lutGH_CD;   /* ground-effect factors over the wing's height above the plane of the tyres' contact points (Wing.cpp:134-139,181-186); n = 0: none */
} pdb_wing;

typedef struct pdb_wing_ctrl {
    int32_t wing;         /* index into wings[] (WING_n first, then FIN_n, as AeroMap builds its list) */
    int32_t input;        /* WingControllerVariable: 1 BRAKE, 2 GAS, 3 LATG, 4 LONG, 5 STEER, 6 SPEED_KMH, 7 / 8 SUS_TRAVEL_LR / RR (such a wing steps after the suspensions of the tick) */
    int32_t combinator;   /* 1 ADD, 2 MULT */
    float filter;         /* ((1 - FILTER) * 1.3333334) * 333.33334 */
    float upLimit, downLimit;
    pdb_curve lut;
} pdb_wing_ctrl;

/* DynamicController (Car/DynamicController.cpp): a chain of stages, each a LUT of a car signal (or a constant) through a first-order filter, added to or
 * multiplied into the running result and clamped when the stage has limits.  The car's controllers share the stage table; a stage's filtered value is a
 * word of the record (pdb_dyn_state.ctrlValue). */
typedef struct pdb_ctrl_stage {
    int32_t input;        /* 1 BRAKE, 2 GAS, 3 LATG, 4 LONG, 5 STEER, 6 SPEED_KMH, 7 GEAR, 8 RPMS, 9 CONST, 10 SLIPRATIO_MAX, 11 SLIPRATIO_AVG, 12 / 13 SLIPANGLE_FRONT / REAR_AVG,
                           * 14 / 15 SLIPANGLE_FRONT / REAR_MAX, 16 OVERSTEER_FACTOR, 17 REAR_SPEED_RATIO, 18 STEER_DEG, 19 WHEEL_STEER_DEG, 20 / 21 LOAD_SPREAD_LF / RF,
                           * 22 AVG_TRAVEL_REAR, 23 / 24 SUS_TRAVEL_LR / RR */
    int32_t combinator;   /* 1 ADD, 2 MULT */
    float filter;         /* lagToLerpDeltaK(FILTER, 0.004, 0.003) */
    float upLimit, downLimit, constValue;
    pdb_curve lut;
} pdb_ctrl_stage;
typedef struct pdb_dyn_ctrl { int32_t first, count; } pdb_dyn_ctrl;
typedef struct pdb_brake_disc { float torqueK, coolTransfer, coolSpeedFactor; pdb_curve perfCurve; } pdb_brake_disc;   /* stages [first, first + count); count 0 = no controller */

typedef struct pdb_scoring {
    float SmoothSteerSpeed, MinBonusSpeed, MaxBonusSpeed, StallRpm, DirectionThreshold, OutOfTrackThreshold,
        ApproachDistance, CriticalDistance, TravelBonus, TravelSplineBonus, DriftBonus, SpeedBonus, ThrottleBonus,
        EngineRpmBonus, DirectionBonus, DirectionPenalty, ObstApproachPenalty, CollisionPenalty, OffTrackPenalty,
        GearGrindPenalty, StallPenalty;
} pdb_scoring;

/* Body colliders of the chassis (Car/CarColliderManager.cpp:12-37 boxes from colliders.ini; Car.cpp:318-377 collider.bin
 * hull).  Vertices are in the chassis body frame (the graphics-offset transform of Car::getGraphicsOffsetMatrix is applied at
 * load time, like addMeshCollider's geom offset).  The box meets TRACK surfaces (C_MASK_CAR_BOX = 1), the hull meets what
 * C_MASK_CAR_MESH = 30 selects (WALL) -- Sim/SimulatorCommon.h:7-13. */
#define PDB_MAX_COLL_VERTS 128
#define PDB_MAX_BOXES 3
#define PDB_MAX_COLL_TRIS 192
typedef struct pdb_collider {
    int32_t enabled;          /* evaluate body contacts (the loader sets 1 when colliders.ini / collider.bin were read) */
    int32_t numBoxes, numVerts, numTris;   /* numBoxes: COLLIDER_0 .. of colliders.ini (CarColliderManager.cpp:17-33 takes every section there is; every shipped car has one) */
    float boxCentre[PDB_MAX_BOXES][3], boxHalf[PDB_MAX_BOXES][3];
    float boundsLo[3], boundsHi[3];   /* body-frame box around hull and belly box: the broad phase works on its world AABB */
    float verts[PDB_MAX_COLL_VERTS][3];
    uint8_t tris[PDB_MAX_COLL_TRIS][3];
} pdb_collider;

typedef struct pdb_car_params {
    int32_t magic;            /* 'PDCP' */
    int32_t version;
    /* rigid-body topology */
    int32_t numBodies, numJoints, numRows;
    pdb_body_def bodies[PDB_MAX_BODIES];
    pdb_joint_def joints[PDB_MAX_JOINTS];
    float worldErp, worldCfm;
    float gravity[3];
    /* car (Car/Car.h:118-147) */
    int32_t suspTypeF, suspTypeR;
    float mass, steerLock, steerRatio, steerLinearRatio, axleTorqueReaction;
    float fuelTankPos[3];
    float fuel, fuelKG, fuelConsumptionK;
    float baseCarHeight;
    float arbK[2];
    float waterTmass, waterCoolSpeedK;
    float probeDir[PDB_NUM_PROBES][3];
    float probeLen[PDB_NUM_PROBES];
    int32_t lookAheadCount;
    float lookAheadStep;
    /* sim (Sim/Simulator.cpp:33-60) */
    float ambientTemperature, roadTemperature, mechanicalDamageRate, tyreConsumptionRate, fuelConsumptionRate;
    float airDensity;          /* Simulator::getAirDensity() */
    pdb_susp susp[4];
    pdb_tyre tyre[4];
    int32_t numWings;
    pdb_wing wings[PDB_MAX_WINGS];
    /* brakes (Car/BrakeSystem.h:44-52) */
    float brakePower, brakePowerMultiplier, handBrakeTorque, frontBias, biasMin, biasMax;
    /* drivetrain (Car/Drivetrain.h:103-121) -- doubles where the reference holds doubles */
    int32_t tractionType, diffType, numGears, isShifterSupported;
    double gearRatio[PDB_MAX_GEARS];
    double finalRatio, diffPowerRamp, diffCoastRamp, diffPreLoad, gearUpTime, gearDnTime, autoCutOffTime;
    double controlsWindowGain, validShiftRPMWindow, damageRpmWindow, clutchMaxTorque, clutchInertia;
    double driveInertia, engineInertiaInit, outShaftInertiaL, outShaftInertiaR;
    /* engine (Car/Engine.h:11-23,58-92) */
    pdb_curve powerCurve, throttleCurve;
    int32_t engMinimum, engLimiter, engLimiterCycles;
    float engCoast1, engCoast2, engInertia, limiterMultiplier, rpmDamageThreshold, rpmDamageK, bovThreshold;
    float maxPowerRPM, maxTorqueRPM;
    pdb_heave heave[2];                  /* front, rear */
    int32_t numTurbos;
    pdb_turbo turbos[PDB_MAX_TURBOS];
    float turboBoostDamageThreshold, turboBoostDamageK;
    int32_t autoTeleport;     /* setCarAutoTeleport (PyProjectD.cpp:292-295; Car.h teleportOnCollision / teleportOnBadLocation / teleportMode):
                               * bit 0 teleport on collision, bit 1 teleport on bad location, bits 2-3 mode (0 Start, 1 Nearest, 2 Random) */
    /* assists */
    float acRpmMin, acRpmMax, acClutchSpeed;
    int32_t acUseOnChange, acUseOnStart, autoShiftActive, autoBlipActive, autoBlipElectronic;
    pdb_curve upshiftProfile, downshiftProfile, blipProfile;
    double blipPerformTime;
    int32_t asChangeUpRpm, asChangeDnRpm;
    float asSlipThreshold, asGasCutoffTime;
    int32_t smoothSteer;
    /* thermal patch connection table (Car/TyreThermalModel.cpp:31-58 build order) */
    int8_t patchConnCount[36];
    int8_t patchConn[36][4];
    pdb_scoring scoring;
    pdb_collider collider;
    /* branches no shipped car takes (round 3; pinned on a derived car, oracle/make_base.py): a second throttle curve blended in by rpm
     * ([THROTTLE_RESPONSE], Engine.cpp:150-154,344-366), the coast offset of [COAST_SETTINGS] (Engine.cpp:61-67,198-207,394-399), the
     * load-proportional brake bias of [EBB] (BrakeSystem.cpp:28-32,92-113), the valve-overlap torque ripple of [OVERLAP] */
    pdb_curve throttleCurveMax;
    float throttleMaxRef;          /* THROTTLE_RESPONSE RPM_REFERENCE */
    float gasCoastOffset;          /* COAST_SETTINGS LUT at its DEFAULT index; 0 = off */
    int32_t coastEntryRpm;         /* engine minimum + ACTIVATION_RPM */
    int32_t ebbInternal;           /* brakes.ini has [EBB] */
    float ebbFrontMultiplier;      /* max(1.1, FRONT_SHARE_MULTIPLIER) */
    float overlapFreq, overlapGain, overlapIdealRPM;   /* [OVERLAP] (Engine.cpp:96-101,300-307): a torque ripple below / above the ideal rpm; gain 0 = off */
    int32_t wingGroundEffect;      /* some wing carries LUT_GH_CL / LUT_GH_CD (Wing.cpp:38-44): the wings then step after the tyres, on this tick's contact points (Car::getPointGroundHeight) */
    /* aero.ini without [WING_n] / [FIN_n]: AeroMap's own drag and lift from [DATA] (AeroMap.cpp:49-58,90-139); CDA is the class default 0.1, the lift acts
     * at the base positions of suspensions 0 and 2 (Car.cpp:176-178) */
    float aeroReferenceArea, aeroFrontShare, aeroCD, aeroCL, aeroCDX, aeroCDY, aeroCDA;
    /* aero.ini [DYNAMIC_CONTROLLER_n] (AeroMap.cpp:66-82, Car/WingDynamicController.cpp): a wing's angle = its ANGLE combined, controller by controller in
     * section order, with a filtered LUT of a car signal and clamped to the controller's limits (Wing.cpp:105-124) */
    int32_t numWingCtrl;
    pdb_wing_ctrl wingCtrl[PDB_MAX_WING_CTRL];
    /* ctrl_single_lock.ini (Drivetrain.cpp:144-151,603-607: the differential's preload, power ramp 0), ctrl_turbo<n>.ini / ctrl_wastegate<n>.ini
     * (Engine.cpp:124-143,368-376: a turbo's maxBoost / wastegate) */
    pdb_dyn_ctrl ctrlDiffLock, ctrlTurboBoost[PDB_MAX_TURBOS], ctrlWastegate[PDB_MAX_TURBOS];
    /* ctrl_ebb.ini (BrakeSystem.cpp:64-69,90-93: the front brake bias of the tick; takes precedence over [EBB]), steer_brake_controller.ini
     * (BrakeSystem.cpp:33-38,136-143: extra brake torque on the inner rear wheel) */
    pdb_dyn_ctrl ctrlEbb, ctrlSteerBrake;
    /* brakes.ini [TEMPS_FRONT] + [TEMPS_REAR] (BrakeSystem.cpp:40-52,151-169): each disc's temperature follows the work done on it and the air stream,
     * and scales its brake torque through PERF_CURVE */
    int32_t hasBrakeTemps;
    float slipEffectGainMult;      /* aero.ini [SLIPSTREAM] EFFECT_GAIN_MULT (AeroMap.cpp:25-29; SlipStream.h default 1): how strongly THIS car's wake thins the air of the cars behind it */
    pdb_brake_disc discs[4];
    /* ctrl_arb_front.ini / ctrl_arb_rear.ini (Car.cpp:158-167, AntirollBar.cpp:19-22): the bar's rate of the tick; the bars step after the drivetrain */
    pdb_dyn_ctrl ctrlArb[2];
    int32_t numCtrlStages;
    float slipSpeedFactorMult;     /* [SLIPSTREAM] SPEED_FACTOR_MULT (default 1): the wake's length = speed x 0.25 x this (Sim/SlipStream.cpp:37-47) */
    pdb_ctrl_stage ctrlStages[PDB_MAX_CTRL_STAGES];
} pdb_car_params;

/* ---------------------------------------------------------------------------------------------
 * Per-car persistent state (everything that lives from one tick to the next)
 * ------------------------------------------------------------------------------------------- */
typedef struct pdb_body_state {
    float pos[3];
    float q[4];
    float R[9];
    float lvel[3];
    float avel[3];
} pdb_body_state;   /* 22 floats */

typedef struct pdb_tyre_state {
    double flatSpot, virtualKM, phase;
    float angularVelocity, slipAngleRAD, slipRatio, ndSlip, load, Fx, Fy, Mz;
    float dirtyLevel, inflation, pressureDynamic, loadedRadius, effectiveRadius, camberRAD, D, localMX, oldAngularVelocity;
    float contactPoint[3], unmodifiedContactPoint[3], contactNormal[3];
    float coreTemp, thermalMultD, practicalTemp;
    float T[36];
    int32_t isLocked;
    float inputT0;   /* TyreThermalPatch::inputT of every patch before the first thermal step (buildTyre sets it to ambient; 0 afterwards) */
} pdb_tyre_state;

typedef struct pdb_dyn_state {
    /* doubles first (8-byte aligned) */
    double physicsTime;
    double engineVel, driveVel, outShaftLVel, outShaftRVel, rootVelocity;
    double gearReqTimeAccumulator, gearReqTimeout, cutOff, lastRatio, validShiftRPMWindow;
    double blipStartTime, fuel;
    double envTotalReward;   /* env mode (pdb_set_env): the episode's cumulative reward, python-float arithmetic (projectd_env.py:199) */
    pdb_body_state body[PDB_MAX_BODIES];
    pdb_tyre_state tyre[4];
    /* car */
    float smoothSteerValue, lastVelocity[3], waterT, lastTrackPointTimestamp, trackLocation, oldTrackLocation;
    float bodyVsTrack, velocityVsTrack, pointCachePos[3], speed;
    int32_t sleepingFrames, nearestTrackPointId, oldTrackPointId, splinePointId;
    /* drivetrain / engine / assists */
    int32_t currentGear, gearReqRequest, gearReqRequestedGear, clutchOpenState, isGearGrinding;
    int32_t limiterOn, lastGearUp, lastGearDn, acSeqActive, acSeqIsDone;
    float lifeLeft, fuelPressure, acClutchValueSignal, acSeqCurrentTime, asGasCutoff;
    /* scoring */
    float totalReward, stepReward, currentDriftAngle, currentSpeedMultiplier, lastDriftDirection, driftStraightTimer;
    float instantDriftDelta, instantDrift, driftPoints;
    int32_t oldPointId, oldSplinePointId, drifting, driftExtreme, driftInvalid, driftComboCounter;
    int32_t collisionFlag, oldCollisionFlag, outOfTrackFlag;
    float gasUsage;          /* Engine::gasUsage of the previous tick (fuel burn input, Car.cpp:478) */
    float locClutch;         /* Drivetrain::locClutch of the previous tick (read by the H-shifter gear select, Drivetrain.cpp:189) */
    float turboRotation[PDB_MAX_TURBOS];   /* Turbo::rotation */
    /* body contacts (PhysicsEngineODE::collisionStep, Car::onCollisionCallback) */
    int32_t simFrame;        /* PhysicsEngineODE::currentFrame: dynamic-vs-static pairs are collided on odd frames only (:230-241) */
    int32_t damageChanged;   /* some damageZoneLevel moved by more than 0.001 this tick (ScoringSystem::validateDrift, :360-368) */
    float damageZoneLevel[5];   /* Car.h:204 */
    int32_t numContacts;     /* contact joints alive in the engine's contactGroupDynamic (the car's row of the pdb_contact array) */
    int32_t randState;       /* the car's C-runtime rand() state (Core/Math.h:49-52 randR -> Car::teleportByMode(Random)); srand(1) at creation */
    int32_t envPending;      /* env mode: 1 = the episode ended on the last tick: the next tick is the reset tick (teleport + step([0,0]),
                              * projectd_env.py:216-227); 2 = reset tick whose teleport the caller has already done (the record stays as the caller left it);
                              * 3 = the episode ended because the record is no longer finite: the next tick re-creates it from a fresh record, then goes on as 1 */
    int32_t envStepId;       /* env mode: ticks since the episode's reset tick (projectd_env.py step_id) */
    int32_t lawTick;         /* pdb_set_law: the row of the law's bias table this car's next write-out takes (wraps at the table's period); 0 and untouched without a law.
                              * Not reference state (it also keeps the record a multiple of 16 bytes: coalesced 16-byte-per-lane copies) */
    float suspTravel[4];     /* ISuspension status.travel of the last suspension step (read by controllers: the brake system's before this tick's step, the others after it) */
    float brakeDiscT[4];     /* BrakeDisc::t (BrakeSystem.cpp:151-169): 0 at creation, the ambient temperature after Car::reset */
    float ctrlValue[PDB_MAX_CTRL_STAGES];   /* DynamicControllerStage::currentValue of the car's controller stages (never reset, as in the reference) */
    float wingCtrlOut[PDB_MAX_WING_CTRL];   /* WingDynamicController::outputAngle of the car's wing controllers (never reset: it outlives Car::reset, as in the reference) */
} pdb_dyn_state;

/* One contact joint between the chassis and the static world (what PhysicsEngineODE::onCollision hands to
 * dJointCreateContact, PhysicsEngineODE.cpp:283-331): created by the collision pass on odd frames, alive in the solve of that
 * tick and of the following even one (the group is emptied only when it is refilled, :228-243).  World coordinates.
 * kind 0: hull vs WALL (surface.mode 28692, mu 0.25, bounce 0.01, soft_cfm 1e-4); kind 1: belly box vs TRACK (mode 28700,
 * mu 0.1, soft_erp 0.714285731, soft_cfm 0.000952380942).  A car keeps its PDB_MAX_CONTACTS deepest contact points. */
#define PDB_MAX_CONTACTS 32
/* A car's slipstream as its last Car::postStep left it (Sim/SlipStream.cpp:37-47 setPosition; Car.cpp:692-694): where the car was, the direction its wake points
 * (against its velocity), the wake's length, the car's EFFECT_GAIN_MULT.  State of a multi-car simulator (pdb_set_world_size): the other cars of the world read it
 * at the top of their next Car::step (Car::updateAirPressure, Car.cpp:557-585).  A new car's is all zero (no wake) until its first tick ends; teleports leave it alone. */
typedef struct pdb_slip_state {
    float pos[3];
    float length;
    float dir[3];
    float effectGainMult;
} pdb_slip_state;

typedef struct pdb_contact {
    float pos[3];
    float depth;
    float normal[3];   /* unit, pointing into the car */
    int32_t kind;
} pdb_contact;

/* per-tick outputs (compact) */
typedef struct pdb_step_out {
    float obs[PDB_OBS_DIM];   /* projectd_env.py:237-275 order */
    float reward;             /* CarState.stepReward; in env mode the env's reward (penalties applied, 0 on a reset tick) */
    int32_t flags;            /* bit0 collisionFlag, bit1 outOfTrackFlag, bit2 stuck (lastTrackPointTimestamp + stuck_timeout < timestamp);
                               * env mode: bit3 terminated (the episode ended on this tick), bit4 this was the episode's reset tick;
                               * bit5 fault: the chassis pose or velocity is not finite (the car needs a reset) */
} pdb_step_out;

/* Per-lane setup and reward weights: what the reference gives every simulator of its own (PyProjectD.cpp:328-365 setCarTune / setScoringVar on that
 * env's car; projectd_env.py:127-132 applies eight tunes and twenty-one scoring variables per env) as a 144-byte row per car of a batch, so that a
 * batch can randomise the setup lane by lane.  The row holds the RESOLVED parameter values -- the fields of pdb_car_params those calls write
 * (Car/SetupManager.cpp: FRONT_BIAS, DIFF_POWER, DIFF_COAST, FINAL_RATIO, PRESSURE_LF/RF/LR/RR; Car/ScoringSystem.cpp:37-101) -- taken from a car
 * block that has been through pdb_set_car_tune / pdb_set_scoring_var (pdb_lane_tune_from_params).  valid = 0: the lane uses its block's values. */
typedef struct pdb_lane_tune {
    double finalRatio, diffPowerRamp, diffCoastRamp;
    float frontBias;
    float pressureStatic[4];
    pdb_scoring scoring;
    int32_t valid;
    int32_t _pad[3];
} pdb_lane_tune;

/* Per-LANE setup, the rest of it (pdb_set_lane_setups): every other SetupManager tune that is a plain field of the car block (Car/SetupManager.cpp:10-120) --
 * BRAKE_POWER_MULT, DIFF_PRELOAD, INTERNAL_GEAR_n, ARB_FRONT/REAR, ENGINE_LIMITER, TURBO_n, and per wheel DAMP_(FAST_)BUMP/REBOUND, BUMP_STOP_RATE, SPRING_RATE,
 * PROGRESSIVE_SPRING_RATE, ROD_LENGTH, PACKER_RANGE, TOE_OUT, CAMBER (as the three entries of the camber rotation the tick uses: DevConst camC/camS/camM33).  The row
 * holds the RESOLVED values of a block that went through pdb_set_car_tune (pdb_lane_setup_from_params).  WING_n is accepted by the reference and overwritten on every
 * tick (csrc/host/model.cpp findTune): nothing to carry. */
typedef struct pdb_lane_wheel {
    float bumpFast, bumpSlow, reboundFast, reboundSlow;
    float bumpStopRate, k, progressiveK, rodLength;
    float packerRange, toeOutLinear, camC, camS;
    float camM33, _pad[3];
} pdb_lane_wheel;
typedef struct pdb_lane_setup {
    double diffPreLoad;
    double gearRatio[PDB_MAX_GEARS];
    float brakePowerMultiplier, limiterMultiplier;
    float arbK[2];
    float turboUserSetting[PDB_MAX_TURBOS];
    float _pad0;
    pdb_lane_wheel wheel[4];
    int32_t _pad1[2];
} pdb_lane_setup;

/* Env mode: the reward / termination / reset bookkeeping of pyprojectd/projectd_env.py:173-227, per car, inside the tick --
 * reward = stepReward minus the penalties of the rules that fired, terminate on hit / off track / stuck / cumulative reward below
 * low_reward; the tick after a termination is the env's reset(): teleport by teleport_mode (if teleport_on_reset) at its top, the
 * zero action, its reward and termination discarded, the episode sums cleared.  One kernel launch = one VecEnv.step().
 * Not in the reference env: a car whose chassis pose or velocity is no longer finite (pdb_step_out.flags bit 5) also ends its episode
 * (reward 0), and at the top of its next tick its record is re-created from a fresh car's before the teleport -- Car::reset cannot repair a
 * record once NaNs are in it (it leaves the struts' velocities, the steering filter, the tyres' loads ... as they are), and a vector env would
 * otherwise carry a dead lane for ever. */
typedef struct pdb_env_config {
    int32_t enabled;
    int32_t terminate_on_hit, terminate_off_track, terminate_when_stuck;
    double hit_penalty, off_track_penalty, stuck_penalty;
    double low_reward;
    int32_t teleport_on_reset, teleport_mode;
} pdb_env_config;

/* ---------------------------------------------------------------------------------------------
 * Track blob: header followed by arrays (offsets in bytes from the start of the blob)
 * ------------------------------------------------------------------------------------------- */
typedef struct pdb_surface {
    float gripMod, damping, sinHeight, sinLength, granularity, dirtAdditiveK;
    int32_t collisionCategory, isValidTrack, triStart, triCount, sectorID, _pad;
} pdb_surface;

typedef struct pdb_track_header {
    int32_t magic;   /* 'PDTK' */
    int32_t version;
    int32_t numSurfaces, numTris, numFat, numNodes;
    int32_t interpolateStep, closedLoop;
    float computedTrackLength, computedTrackWidth, dynamicGripLevel, hashCellSize;
    uint64_t offSurfaces;   /* pdb_surface[numSurfaces] */
    uint64_t offTris;       /* float[numTris][9]  (v0,v1,v2) */
    uint64_t offFat;        /* float[numFat][15]  best,left,right,center,forwardDir (Sim/Track.h:18-25) */
    uint64_t offFatDist;    /* float[numFat] */
    uint64_t offNodes;      /* float[numNodes][3] interpolated B-spline nodes */
    uint64_t offNodeDist;   /* float[numNodes] */
    uint64_t totalBytes;
    /* version >= 2: uniform xz grid over the surface triangles for the (vertical) wheel rays.  Cell (ix, iz) lists, in
     * ascending triangle order, every triangle whose xz bounding box overlaps it; a vertical ray at (x, z) needs only the
     * list of its own cell and finds exactly the hits of a scan over all triangles. */
    int32_t gridNx, gridNz;
    float gridMinX, gridMinZ, gridCell, _gridPad;
    uint64_t offGridStart;  /* int32[gridNx*gridNz + 1] */
    uint64_t offGridTris;   /* int32[...]  triangle ids */
    uint64_t offTriSurf;    /* int32[numTris] surface of each triangle (low 24 bits) | its collisionCategory << 24 (version >= 4) */
    /* version >= 3: uniform xz grid over the fat points (cell edge = hashCellSize, the reference's VertexHash cell,
     * Core/VertexHash.h:45-104) for Track::nearbyPoints: cell (ix, iz) lists its points in ascending id; cells are
     * stored row by row, so the lists of the cells ix0..ix1 of one row are one contiguous run of ids. */
    int32_t fatGridNx, fatGridNz;
    float fatGridMinX, fatGridMinZ, fatGridCell, _fatGridPad;
    uint64_t offFatGridStart;   /* int32[fatGridNx*fatGridNz + 1] */
    uint64_t offFatGridIds;     /* int32[numFat] */
    /* version >= 4: the wheel rays' own grid.  Cell (ix, iz) lists, in ascending triangle order, a COPY of every triangle whose
     * xz projection (not just its bounding box) can contain a point of the cell, so that the candidates of a ray are one
     * contiguous run of records behind two dependent loads, and a finer cell than the collision grid's costs no accuracy. */
    int32_t rayNx, rayNz;
    float rayMinX, rayMinZ, rayCell, _rayPad;
    uint64_t offRayStart;   /* int32[rayNx*rayNz + 1] */
    uint64_t offRayRecs;    /* pdb_ray_rec[...] */
    /* version >= 5: what the probe scans read, laid out for one load per candidate.  offFatGridRec: parallel to offFatGridIds,
     * float[numFat][4] = the listed point's best position and, in the fourth word, its id (int32 bits).  offFatSeg:
     * float[numFat][8] = the two side segments leaving point i: left(i).xz, left(i+1).xz, right(i).xz, right(i+1).xz
     * (i+1 wraps to 0, Track.cpp:520-560). */
    uint64_t offFatGridRec;
    uint64_t offFatSeg;
    /* version >= 6: the pit boxes of pits.ini (Track::loadPits, Sim/Track.cpp:151-175): float[numPits][16], the reference's mat44f row by row
     * (M11..M44) = createFromAxisAngle((0,1,0), ROT.x degrees) with POS in M41..M43.  Car::teleportToPits reads the heading (M31..M33) and the
     * position (M41..M43) of pit `id` (Car.cpp:1310-1323). */
    int32_t numPits, _pitPad;
    uint64_t offPits;
} pdb_track_header;

typedef struct pdb_ray_rec {
    float v[9];         /* v0, v1, v2 as in offTris */
    int32_t tri;        /* triangle id */
    int32_t surface;
    int32_t _pad;
} pdb_ray_rec;

#ifdef __cplusplus
}
/* sizes are part of the ABI (tests/test_abi.py reads these numbers) */
static_assert(sizeof(pdb_car_state) == 664, "pdb_car_state must equal the reference CarState (pack 4)");
static_assert(sizeof(pdb_car_params) == 22696, "pdb_car_params layout");
static_assert(sizeof(pdb_dyn_state) == 2352, "pdb_dyn_state layout (multiple of 16 bytes)");
static_assert(sizeof(pdb_step_out) == 104, "pdb_step_out layout");
static_assert(sizeof(pdb_contact) == 32, "pdb_contact layout");
static_assert(sizeof(pdb_slip_state) == 32, "pdb_slip_state layout");
static_assert(sizeof(pdb_lane_setup) == 384 && sizeof(pdb_lane_wheel) == 64, "pdb_lane_setup layout");
static_assert(sizeof(pdb_lane_tune) == 144, "pdb_lane_tune layout (a multiple of 16 bytes: it rides in the car's LDS block)");
static_assert(sizeof(pdb_ray_rec) == 48, "pdb_ray_rec layout");
#endif
#endif
