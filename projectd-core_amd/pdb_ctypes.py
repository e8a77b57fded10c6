"""ctypes views of include/pdb_types.h and include/pdbatch.h (the product library only)."""
import ctypes as C, os, sys
import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)

MAX_CURVE, MAX_BODIES, MAX_JOINTS, MAX_WINGS, MAX_GEARS = 24, 8, 24, 6, 10

class Curve(C.Structure):
    _fields_ = [('n', C.c_int32), ('x', C.c_float * MAX_CURVE), ('y', C.c_float * MAX_CURVE)]
class Controls(C.Structure):
    _pack_ = 4
    _fields_ = [('steer', C.c_float), ('clutch', C.c_float), ('brake', C.c_float), ('handBrake', C.c_float), ('gas', C.c_float),
                ('isShifterSupported', C.c_int8), ('requestedGearIndex', C.c_int8), ('gearUp', C.c_int8), ('gearDn', C.c_int8)]
class CarState(C.Structure):
    _pack_ = 4
    _fields_ = [('carId', C.c_int32), ('simId', C.c_int32), ('timestamp', C.c_float), ('controls', Controls),
                ('collisionFlag', C.c_int32), ('outOfTrackFlag', C.c_int32), ('trackPointId', C.c_int32),
                ('lastTrackPointTimestamp', C.c_float), ('trackLocation', C.c_float), ('bodyVsTrack', C.c_float), ('velocityVsTrack', C.c_float),
                ('engineRPM', C.c_float), ('speedMS', C.c_float), ('gear', C.c_int32), ('gearGrinding', C.c_int32),
                ('bodyMatrix', C.c_float * 16), ('bodyPos', C.c_float * 3), ('bodyEuler', C.c_float * 3), ('accG', C.c_float * 3),
                ('velocity', C.c_float * 3), ('localVelocity', C.c_float * 3), ('angularVelocity', C.c_float * 3), ('localAngularVelocity', C.c_float * 3),
                ('hubMatrix', (C.c_float * 16) * 4), ('tyreContacts', (C.c_float * 3) * 4),
                ('tyreLoad', C.c_float * 4), ('tyreAngularSpeed', C.c_float * 4), ('tyreSlipRatio', C.c_float * 4), ('tyreNdSlip', C.c_float * 4),
                ('probes', C.c_float * 10), ('lookAhead', C.c_float * 5), ('stepReward', C.c_float), ('totalReward', C.c_float)]
assert C.sizeof(CarState) == 664
class BodyDef(C.Structure):
    _fields_ = [('mass', C.c_float), ('inertia', C.c_float * 3)]
class JointDef(C.Structure):
    _fields_ = [('type', C.c_int32), ('b0', C.c_int32), ('b1', C.c_int32), ('suspErp', C.c_int32), ('steerWheel', C.c_int32),
                ('erp', C.c_float), ('cfm', C.c_float), ('anchor1', C.c_float * 3), ('anchor2', C.c_float * 3), ('axis1', C.c_float * 3),
                ('offset', C.c_float * 3), ('qrel', C.c_float * 4), ('distance', C.c_float)]
class Damper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('bumpSlow', 'reboundSlow', 'bumpFast', 'reboundFast', 'fastThresholdBump', 'fastThresholdRebound')]
class Susp(C.Structure):
    _fields_ = [('type', C.c_int32), ('hubBody', C.c_int32), ('strutBody', C.c_int32)] + \
        [(n, C.c_float) for n in ('k', 'progressiveK', 'bumpStopUp', 'bumpStopDn', 'bumpStopRate', 'rodLength', 'toeOutLinear', 'staticCamber', 'packerRange')] + \
        [('damper', Damper), ('basePosition', C.c_float * 3), ('carStrut', C.c_float * 3), ('tyreStrut', C.c_float * 3), ('tyreSteer', C.c_float * 3),
         ('baseCarSteer', C.c_float * 3), ('refPointY', C.c_float), ('refPointSignX', C.c_float), ('strutBaseLength', C.c_float), ('strutBodyLength', C.c_float),
         ('axleTrack', C.c_float), ('referenceY', C.c_float), ('attachRelativePos', C.c_float), ('leafSpringKx', C.c_float), ('axleBasePos', C.c_float * 3),
         ('sideSign', C.c_float), ('mass', C.c_float), ('bumpStopProgressive', C.c_float)]
class Heave(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('k', 'progressiveK', 'bumpStopUp', 'bumpStopDn', 'rodLength', 'bumpStopRate', 'packerRange')] + [('damper', Damper), ('_pad', C.c_int32)]
class Turbo(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('lagDN', 'lagUP', 'maxBoost', 'wastegate', 'rpmRef', 'gamma', 'userSetting')] + [('isAdjustable', C.c_int32)]
class Collider(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('enabled', 'numBoxes', 'numVerts', 'numTris')] + \
        [('boxCentre', (C.c_float * 3) * 3), ('boxHalf', (C.c_float * 3) * 3), ('boundsLo', C.c_float * 3), ('boundsHi', C.c_float * 3),
         ('verts', (C.c_float * 3) * 128), ('tris', (C.c_uint8 * 3) * 192)]
class Spline(C.Structure):
    _fields_ = [('n', C.c_int32), ('b0', C.c_float), ('c0', C.c_float), ('pad', C.c_int32)] + [(k, C.c_float * 16) for k in ('x', 'y', 'a', 'b', 'c')]
class Tyre(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('radius', 'rimRadius', 'k', 'd', 'angularInertia', 'thermalFrictionK', 'thermalRollingK', 'thermalRollingSurfaceK',
                                         'radiusRaiseK', 'softnessIndex', 'Fz0', 'modelFz0', 'relaxationLength', 'rr0', 'rr1', 'rr_slip', 'pressureRef', 'pressureSpringGain',
                                         'pressureRRGain', 'pressureGainD', 'idealPressure', 'flatSpotK', 'explosionTemperature', 'pressureTemperatureGain', 'pressureStatic')] + \
        [('version', C.c_int32), ('driven', C.c_int32)] + \
        [(n, C.c_float) for n in ('lsMultY', 'lsExpY', 'lsMultX', 'lsExpX', 'maxSlip0', 'maxSlip1', 'asy', 'falloffSpeed', 'speedSensitivity', 'camberGain',
                                  'dcamber0', 'dcamber1', 'cfXmult', 'pressureCfGain', 'brakeDXMod', 'dCamberBlend', 'combinedFactor',
                                  'surfaceTransfer', 'patchTransfer', 'patchCoreTransfer', 'internalCoreTransfer', 'coolFactorGain', 'camberSpreadK')] + \
        [('performanceCurve', Curve), ('wearCurve', Curve), ('curveFlags', C.c_int32), ('curvePad', C.c_int32), ('dyLoadCurve', Spline), ('dxLoadCurve', Spline), ('dCamberCurve', Spline)]
class WingCtrl(C.Structure):
    _fields_ = [('wing', C.c_int32), ('input', C.c_int32), ('combinator', C.c_int32), ('filter', C.c_float), ('upLimit', C.c_float), ('downLimit', C.c_float), ('lut', Curve)]
class CtrlStage(C.Structure):
    _fields_ = [('input', C.c_int32), ('combinator', C.c_int32), ('filter', C.c_float), ('upLimit', C.c_float), ('downLimit', C.c_float), ('constValue', C.c_float), ('lut', Curve)]
class BrakeDisc(C.Structure):
    _fields_ = [('torqueK', C.c_float), ('coolTransfer', C.c_float), ('coolSpeedFactor', C.c_float), ('perfCurve', Curve)]
class DynCtrl(C.Structure):
    _fields_ = [('first', C.c_int32), ('count', C.c_int32)]
class Wing(C.Structure):
    _fields_ = [('position', C.c_float * 3), ('area', C.c_float), ('cdGain', C.c_float), ('clGain', C.c_float), ('yawGain', C.c_float), ('angle', C.c_float),
                ('isVertical', C.c_int32), ('lutAOA_CL', Curve), ('lutAOA_CD', Curve), ('lutGH_CL', Curve), ('lutGH_CD', Curve)]
SCORING_VARS = ['SmoothSteerSpeed', 'MinBonusSpeed', 'MaxBonusSpeed', 'StallRpm', 'DirectionThreshold', 'OutOfTrackThreshold', 'ApproachDistance',
                'CriticalDistance', 'TravelBonus', 'TravelSplineBonus', 'DriftBonus', 'SpeedBonus', 'ThrottleBonus', 'EngineRpmBonus', 'DirectionBonus',
                'DirectionPenalty', 'ObstApproachPenalty', 'CollisionPenalty', 'OffTrackPenalty', 'GearGrindPenalty', 'StallPenalty']
class Scoring(C.Structure):
    _fields_ = [(n, C.c_float) for n in SCORING_VARS]
class CarParams(C.Structure):
    _fields_ = [('magic', C.c_int32), ('version', C.c_int32), ('numBodies', C.c_int32), ('numJoints', C.c_int32), ('numRows', C.c_int32),
                ('bodies', BodyDef * MAX_BODIES), ('joints', JointDef * MAX_JOINTS), ('worldErp', C.c_float), ('worldCfm', C.c_float), ('gravity', C.c_float * 3),
                ('suspTypeF', C.c_int32), ('suspTypeR', C.c_int32), ('mass', C.c_float), ('steerLock', C.c_float), ('steerRatio', C.c_float),
                ('steerLinearRatio', C.c_float), ('axleTorqueReaction', C.c_float), ('fuelTankPos', C.c_float * 3), ('fuel', C.c_float), ('fuelKG', C.c_float),
                ('fuelConsumptionK', C.c_float), ('baseCarHeight', C.c_float), ('arbK', C.c_float * 2), ('waterTmass', C.c_float), ('waterCoolSpeedK', C.c_float),
                ('probeDir', (C.c_float * 3) * 7), ('probeLen', C.c_float * 7), ('lookAheadCount', C.c_int32), ('lookAheadStep', C.c_float),
                ('ambientTemperature', C.c_float), ('roadTemperature', C.c_float), ('mechanicalDamageRate', C.c_float), ('tyreConsumptionRate', C.c_float),
                ('fuelConsumptionRate', C.c_float), ('airDensity', C.c_float), ('susp', Susp * 4), ('tyre', Tyre * 4), ('numWings', C.c_int32), ('wings', Wing * MAX_WINGS),
                ('brakePower', C.c_float), ('brakePowerMultiplier', C.c_float), ('handBrakeTorque', C.c_float), ('frontBias', C.c_float), ('biasMin', C.c_float), ('biasMax', C.c_float),
                ('tractionType', C.c_int32), ('diffType', C.c_int32), ('numGears', C.c_int32), ('isShifterSupported', C.c_int32), ('gearRatio', C.c_double * MAX_GEARS)] + \
        [(n, C.c_double) for n in ('finalRatio', 'diffPowerRamp', 'diffCoastRamp', 'diffPreLoad', 'gearUpTime', 'gearDnTime', 'autoCutOffTime', 'controlsWindowGain',
                                   'validShiftRPMWindow', 'damageRpmWindow', 'clutchMaxTorque', 'clutchInertia', 'driveInertia', 'engineInertiaInit', 'outShaftInertiaL', 'outShaftInertiaR')] + \
        [('powerCurve', Curve), ('throttleCurve', Curve), ('engMinimum', C.c_int32), ('engLimiter', C.c_int32), ('engLimiterCycles', C.c_int32)] + \
        [(n, C.c_float) for n in ('engCoast1', 'engCoast2', 'engInertia', 'limiterMultiplier', 'rpmDamageThreshold', 'rpmDamageK', 'bovThreshold', 'maxPowerRPM', 'maxTorqueRPM')] + \
        [('heave', Heave * 2), ('numTurbos', C.c_int32), ('turbos', Turbo * 3), ('turboBoostDamageThreshold', C.c_float), ('turboBoostDamageK', C.c_float), ('autoTeleport', C.c_int32)] + \
        [(n, C.c_float) for n in ('acRpmMin', 'acRpmMax', 'acClutchSpeed')] + \
        [(n, C.c_int32) for n in ('acUseOnChange', 'acUseOnStart', 'autoShiftActive', 'autoBlipActive', 'autoBlipElectronic')] + \
        [('upshiftProfile', Curve), ('downshiftProfile', Curve), ('blipProfile', Curve), ('blipPerformTime', C.c_double), ('asChangeUpRpm', C.c_int32), ('asChangeDnRpm', C.c_int32),
         ('asSlipThreshold', C.c_float), ('asGasCutoffTime', C.c_float), ('smoothSteer', C.c_int32), ('patchConnCount', C.c_int8 * 36), ('patchConn', (C.c_int8 * 4) * 36),
         ('scoring', Scoring), ('collider', Collider), ('throttleCurveMax', Curve), ('throttleMaxRef', C.c_float), ('gasCoastOffset', C.c_float),
         ('coastEntryRpm', C.c_int32), ('ebbInternal', C.c_int32), ('ebbFrontMultiplier', C.c_float), ('overlapFreq', C.c_float), ('overlapGain', C.c_float), ('overlapIdealRPM', C.c_float), ('wingGroundEffect', C.c_int32)] + [(k, C.c_float) for k in ('aeroReferenceArea', 'aeroFrontShare', 'aeroCD', 'aeroCL', 'aeroCDX', 'aeroCDY', 'aeroCDA')] + [('numWingCtrl', C.c_int32), ('wingCtrl', WingCtrl * 4), ('ctrlDiffLock', DynCtrl), ('ctrlTurboBoost', DynCtrl * 3), ('ctrlWastegate', DynCtrl * 3), ('ctrlEbb', DynCtrl), ('ctrlSteerBrake', DynCtrl), ('hasBrakeTemps', C.c_int32), ('slipEffectGainMult', C.c_float), ('discs', BrakeDisc * 4), ('ctrlArb', DynCtrl * 2), ('numCtrlStages', C.c_int32), ('slipSpeedFactorMult', C.c_float), ('ctrlStages', CtrlStage * 8)]
class BodyState(C.Structure):
    _fields_ = [('pos', C.c_float * 3), ('q', C.c_float * 4), ('R', C.c_float * 9), ('lvel', C.c_float * 3), ('avel', C.c_float * 3)]
class TyreState(C.Structure):
    _fields_ = [('flatSpot', C.c_double), ('virtualKM', C.c_double), ('phase', C.c_double)] + \
        [(n, C.c_float) for n in ('angularVelocity', 'slipAngleRAD', 'slipRatio', 'ndSlip', 'load', 'Fx', 'Fy', 'Mz', 'dirtyLevel', 'inflation', 'pressureDynamic',
                                  'loadedRadius', 'effectiveRadius', 'camberRAD', 'D', 'localMX', 'oldAngularVelocity')] + \
        [('contactPoint', C.c_float * 3), ('unmodifiedContactPoint', C.c_float * 3), ('contactNormal', C.c_float * 3), ('coreTemp', C.c_float), ('thermalMultD', C.c_float),
         ('practicalTemp', C.c_float), ('T', C.c_float * 36), ('isLocked', C.c_int32), ('inputT0', C.c_float)]
class DynState(C.Structure):
    _fields_ = [(n, C.c_double) for n in ('physicsTime', 'engineVel', 'driveVel', 'outShaftLVel', 'outShaftRVel', 'rootVelocity', 'gearReqTimeAccumulator', 'gearReqTimeout',
                                          'cutOff', 'lastRatio', 'validShiftRPMWindow', 'blipStartTime', 'fuel', 'envTotalReward')] + \
        [('body', BodyState * MAX_BODIES), ('tyre', TyreState * 4), ('smoothSteerValue', C.c_float), ('lastVelocity', C.c_float * 3), ('waterT', C.c_float),
         ('lastTrackPointTimestamp', C.c_float), ('trackLocation', C.c_float), ('oldTrackLocation', C.c_float), ('bodyVsTrack', C.c_float), ('velocityVsTrack', C.c_float),
         ('pointCachePos', C.c_float * 3), ('speed', C.c_float)] + \
        [(n, C.c_int32) for n in ('sleepingFrames', 'nearestTrackPointId', 'oldTrackPointId', 'splinePointId', 'currentGear', 'gearReqRequest', 'gearReqRequestedGear',
                                  'clutchOpenState', 'isGearGrinding', 'limiterOn', 'lastGearUp', 'lastGearDn', 'acSeqActive', 'acSeqIsDone')] + \
        [(n, C.c_float) for n in ('lifeLeft', 'fuelPressure', 'acClutchValueSignal', 'acSeqCurrentTime', 'asGasCutoff', 'totalReward', 'stepReward', 'currentDriftAngle',
                                  'currentSpeedMultiplier', 'lastDriftDirection', 'driftStraightTimer', 'instantDriftDelta', 'instantDrift', 'driftPoints')] + \
        [(n, C.c_int32) for n in ('oldPointId', 'oldSplinePointId', 'drifting', 'driftExtreme', 'driftInvalid', 'driftComboCounter', 'collisionFlag', 'oldCollisionFlag',
                                  'outOfTrackFlag')] + \
        [('gasUsage', C.c_float), ('locClutch', C.c_float), ('turboRotation', C.c_float * 3), ('simFrame', C.c_int32), ('damageChanged', C.c_int32),
         ('damageZoneLevel', C.c_float * 5), ('numContacts', C.c_int32), ('randState', C.c_int32), ('envPending', C.c_int32), ('envStepId', C.c_int32), ('lawTick', C.c_int32), ('suspTravel', C.c_float * 4), ('brakeDiscT', C.c_float * 4), ('ctrlValue', C.c_float * 8), ('wingCtrlOut', C.c_float * 4)]
assert C.sizeof(DynState) % 16 == 0
MAX_CONTACTS = 32
class Contact(C.Structure):   # pdb_contact
    _fields_ = [('pos', C.c_float * 3), ('depth', C.c_float), ('normal', C.c_float * 3), ('kind', C.c_int32)]
assert C.sizeof(Contact) == 32
class EnvConfig(C.Structure):   # pdb_env_config
    _fields_ = [('enabled', C.c_int32), ('terminate_on_hit', C.c_int32), ('terminate_off_track', C.c_int32), ('terminate_when_stuck', C.c_int32),
                ('hit_penalty', C.c_double), ('off_track_penalty', C.c_double), ('stuck_penalty', C.c_double), ('low_reward', C.c_double),
                ('teleport_on_reset', C.c_int32), ('teleport_mode', C.c_int32)]
class LaneTune(C.Structure):   # pdb_lane_tune
    _fields_ = [('finalRatio', C.c_double), ('diffPowerRamp', C.c_double), ('diffCoastRamp', C.c_double), ('frontBias', C.c_float), ('pressureStatic', C.c_float * 4),
                ('scoring', Scoring), ('valid', C.c_int32), ('_pad', C.c_int32 * 3)]
assert C.sizeof(LaneTune) == 144
class LaneWheel(C.Structure):   # pdb_lane_wheel
    _fields_ = [(n, C.c_float) for n in ('bumpFast', 'bumpSlow', 'reboundFast', 'reboundSlow', 'bumpStopRate', 'k', 'progressiveK', 'rodLength', 'packerRange', 'toeOutLinear',
                                         'camC', 'camS', 'camM33')] + [('_pad', C.c_float * 3)]
class LaneSetup(C.Structure):   # pdb_lane_setup
    _fields_ = [('diffPreLoad', C.c_double), ('gearRatio', C.c_double * MAX_GEARS), ('brakePowerMultiplier', C.c_float), ('limiterMultiplier', C.c_float), ('arbK', C.c_float * 2),
                ('turboUserSetting', C.c_float * 3), ('_pad0', C.c_float), ('wheel', LaneWheel * 4), ('_pad1', C.c_int32 * 2)]
assert C.sizeof(LaneSetup) == 384
class Surface(C.Structure):   # pdb_surface
    _fields_ = [(n, C.c_float) for n in ('gripMod', 'damping', 'sinHeight', 'sinLength', 'granularity', 'dirtAdditiveK')] + \
               [(n, C.c_int32) for n in ('collisionCategory', 'isValidTrack', 'triStart', 'triCount', 'sectorID', '_pad')]
class TrackHeader(C.Structure):   # pdb_track_header (version 6)
    _fields_ = [(n, C.c_int32) for n in ('magic', 'version', 'numSurfaces', 'numTris', 'numFat', 'numNodes', 'interpolateStep', 'closedLoop')] + \
               [(n, C.c_float) for n in ('computedTrackLength', 'computedTrackWidth', 'dynamicGripLevel', 'hashCellSize')] + \
               [(n, C.c_uint64) for n in ('offSurfaces', 'offTris', 'offFat', 'offFatDist', 'offNodes', 'offNodeDist', 'totalBytes')] + \
               [('gridNx', C.c_int32), ('gridNz', C.c_int32), ('gridMinX', C.c_float), ('gridMinZ', C.c_float), ('gridCell', C.c_float), ('_gridPad', C.c_float)] + \
               [(n, C.c_uint64) for n in ('offGridStart', 'offGridTris', 'offTriSurf')] + \
               [('fatGridNx', C.c_int32), ('fatGridNz', C.c_int32), ('fatGridMinX', C.c_float), ('fatGridMinZ', C.c_float), ('fatGridCell', C.c_float), ('_fatGridPad', C.c_float)] + \
               [(n, C.c_uint64) for n in ('offFatGridStart', 'offFatGridIds')] + \
               [('rayNx', C.c_int32), ('rayNz', C.c_int32), ('rayMinX', C.c_float), ('rayMinZ', C.c_float), ('rayCell', C.c_float), ('_rayPad', C.c_float)] + \
               [(n, C.c_uint64) for n in ('offRayStart', 'offRayRecs', 'offFatGridRec', 'offFatSeg')] + \
               [('numPits', C.c_int32), ('_pitPad', C.c_int32), ('offPits', C.c_uint64)]
class SlipState(C.Structure):   # pdb_slip_state
    _fields_ = [('pos', C.c_float * 3), ('length', C.c_float), ('dir', C.c_float * 3), ('effectGainMult', C.c_float)]
class StepOut(C.Structure):
    _fields_ = [('obs', C.c_float * 24), ('reward', C.c_float), ('flags', C.c_int32)]

def _load(path):
    if not os.path.exists(path):
        raise RuntimeError('%s not built (run: python -c "import __graft_entry__ as g; g.build()")' % path)
    return C.CDLL(path)

def load_product(host_only=False):
    # PDB_LIB: experiment hook (tools/): load a differently-built variant of the device library
    lib = _load(os.path.join(PKG, 'libpdbhost.so' if host_only else os.environ.get('PDB_LIB', 'libpdbatch.so')))
    lib.pdb_last_error.restype = C.c_char_p
    lib.pdb_version.restype = C.c_char_p
    lib.pdb_get_scoring_var.restype = C.c_float
    lib.pdb_get_scoring_var.argtypes = [C.c_void_p, C.c_char_p]
    lib.pdb_set_car_tune.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_float, C.c_int]
    lib.pdb_set_scoring_var.argtypes = [C.c_void_p, C.c_char_p, C.c_float]
    lib.pdb_teleport_to_spline.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.pdb_teleport_by_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    if not host_only and hasattr(lib, 'pdb_set_world_size'):   # (a library variant built before round 6 has none: tools/ A/B runs)
        lib.pdb_set_world_size.argtypes = [C.c_void_p, C.c_int]
        lib.pdb_get_slipstreams.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]; lib.pdb_set_slipstreams.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pdb_teleport_to_pit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.pdb_teleport_to_location.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    lib.pdb_track_num_pits.argtypes = [C.c_void_p]
    lib.pdb_track_pit.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.pdb_set_auto_teleport.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.pdb_lane_tune_from_params.argtypes = [C.c_void_p, C.c_void_p]
    lib.pdb_lane_setup_from_params.argtypes = [C.c_void_p, C.c_void_p]
    if not host_only:
        lib.pdb_create.restype = C.c_void_p
        lib.pdb_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        lib.pdb_destroy.argtypes = [C.c_void_p]
        lib.pdb_num_cars.argtypes = [C.c_void_p]
        lib.pdb_set_state_all.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pdb_get_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pdb_get_contacts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pdb_set_contacts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pdb_reset.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_reset_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.pdb_reset_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.pdb_reset_mask_device.restype = C.c_void_p; lib.pdb_reset_mask_device.argtypes = [C.c_void_p]
        lib.pdb_set_stuck_timeout.argtypes = [C.c_void_p, C.c_double]
        lib.pdb_set_seed.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_set_env.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_set_partition_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(lib, 'pdb_set_contact_grid'):
            lib.pdb_set_contact_grid.argtypes = [C.c_void_p, C.c_int]
        if hasattr(lib, 'pdb_set_lane_tunes'):
            lib.pdb_set_lane_tunes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        if hasattr(lib, 'pdb_set_law'):
            lib.pdb_set_law.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        if hasattr(lib, 'pdb_set_lane_setups'):
            lib.pdb_set_lane_setups.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pdb_actions_device.restype = C.c_void_p; lib.pdb_actions_device.argtypes = [C.c_void_p]
        lib.pdb_out_device.restype = C.c_void_p; lib.pdb_out_device.argtypes = [C.c_void_p]
        lib.pdb_set_out_device.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_stream.restype = C.c_void_p; lib.pdb_stream.argtypes = [C.c_void_p]
        lib.pdb_step.argtypes = [C.c_void_p, C.c_float]
        lib.pdb_step_n.argtypes = [C.c_void_p, C.c_float, C.c_int]
        lib.pdb_sync.argtypes = [C.c_void_p]
        lib.pdb_step_async.argtypes = [C.c_void_p, C.c_float]
        lib.pdb_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_event_record.argtypes = [C.c_void_p, C.c_int]
        lib.pdb_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_step_host.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        lib.pdb_set_partitions.argtypes = [C.c_void_p, C.c_int]
        lib.pdb_step_ring.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.pdb_wait_partitions.argtypes = [C.c_void_p, C.c_void_p]
        lib.pdb_step_partition.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        if 'PDB_LIB' not in os.environ or hasattr(lib, 'pdb_contact_pass_load'):   # (an older variant under the experiment hook lacks these)
            lib.pdb_contact_pass_load.argtypes = [C.c_void_p, C.c_int]
            lib.pdb_clear_episodes.argtypes = [C.c_void_p, C.c_void_p]
            lib.pdb_host_actions.restype = C.c_void_p; lib.pdb_host_actions.argtypes = [C.c_void_p]
            lib.pdb_host_out.restype = C.c_void_p; lib.pdb_host_out.argtypes = [C.c_void_p]
            lib.pdb_step_host_partition.argtypes = [C.c_void_p, C.c_float, C.c_int]
            lib.pdb_wait_host_partition.argtypes = [C.c_void_p, C.c_int]
        lib.pdb_partition_stream.restype = C.c_void_p; lib.pdb_partition_stream.argtypes = [C.c_void_p, C.c_int]
        if hasattr(lib, 'pdb_comm_init'):
            lib.pdb_comm_unique_id.argtypes = [C.c_void_p]
            lib.pdb_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
            lib.pdb_comm_destroy.argtypes = [C.c_void_p]
            lib.pdb_step_exchange_partition.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        lib.pdb_partition_range.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.pdb_partition_mark.argtypes = [C.c_void_p]
        lib.pdb_partition_elapsed_ms.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.pdb_get_car_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.pdb_kernel_time_us.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(lib, 'pdb_sample_kernel'):
            lib.pdb_sample_kernel.argtypes = [C.c_void_p, C.c_int]
            lib.pdb_sampled_kernel_us.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib

ENV_TUNES = {'FRONT_BIAS': 55.0, 'DIFF_POWER': 30.0, 'DIFF_COAST': 30.0, 'FINAL_RATIO': 5.0,
             'PRESSURE_LF': 28.0, 'PRESSURE_RF': 28.0, 'PRESSURE_LR': 28.0, 'PRESSURE_RR': 28.0}
ENV_SCORING = {'SmoothSteerSpeed': 10.0, 'MinBonusSpeed': 5.0, 'MaxBonusSpeed': 200.0, 'StallRpm': 300.0, 'DirectionThreshold': 0.75,
               'OutOfTrackThreshold': 0.51, 'ApproachDistance': 3.5, 'CriticalDistance': 2.0, 'TravelBonus': 0.1, 'TravelSplineBonus': 0.01,
               'DriftBonus': 0.0, 'SpeedBonus': 0.0, 'ThrottleBonus': 0.0, 'EngineRpmBonus': 0.0, 'DirectionBonus': 0.0, 'DirectionPenalty': 0.0,
               'ObstApproachPenalty': 0.0, 'CollisionPenalty': 0.0, 'OffTrackPenalty': 0.0, 'GearGrindPenalty': 0.0, 'StallPenalty': 0.0}

def env_params(lib, base, model='ks_toyota_ae86_drift'):
    """pdb_car_params configured like pyprojectd/projectd_env.py:118-136 (tunes, assists, scoring vars)."""
    P = CarParams()
    rc = lib.pdb_build_car_model(base.encode(), model.encode(), C.byref(P))
    if rc != 0:
        raise RuntimeError(lib.pdb_last_error().decode())
    lib.pdb_set_assists(C.byref(P), 1, 1, 1, 1)
    if model == 'ks_toyota_ae86_drift':
        for k, v in ENV_TUNES.items():
            lib.pdb_set_car_tune(C.byref(P), base.encode(), model.encode(), k.encode(), v, 0)
    for k, v in ENV_SCORING.items():
        assert lib.pdb_set_scoring_var(C.byref(P), k.encode(), v) == 0
    return P

def build_track(lib, base, name, recompute_fat_points=False):
    blob = C.c_void_p(); n = C.c_uint64()
    rc = lib.pdb_build_track_opts(base.encode(), name.encode(), 1 if recompute_fat_points else 0, C.byref(blob), C.byref(n))
    if rc != 0:
        raise RuntimeError(lib.pdb_last_error().decode())
    data = C.string_at(blob, n.value)
    lib.pdb_free(blob)
    return data


# ---- packed tracks: the product's track blob, compressed, for machines without the reference's content/ directory (tools/pack_tracks.py)
TRACK_PACK_DIR = os.path.join(PKG, 'data', 'tracks')
_PACK_MAGIC = b'PDTZ0001'

def track_pack_path(name):
    return os.path.join(TRACK_PACK_DIR, name + '.pdtrack.z')

def write_track_pack(path, blob):
    import zlib, struct
    tmp = path + '.tmp%d' % os.getpid()
    with open(tmp, 'wb') as f:
        f.write(_PACK_MAGIC + struct.pack('<QI', len(blob), zlib.crc32(blob)) + zlib.compress(blob, 6))
    os.replace(tmp, path)

def load_track_pack(name):
    """the blob pdb_build_track gave for this track in the build container; RuntimeError when the pack is absent"""
    import zlib, struct
    path = track_pack_path(name)
    if not os.path.exists(path):
        raise RuntimeError('%s not packed (build container: python tools/pack_tracks.py)' % path)
    raw = open(path, 'rb').read()
    if raw[:8] != _PACK_MAGIC:
        raise RuntimeError('%s: not a track pack' % path)
    n, crc = struct.unpack('<QI', raw[8:20])
    blob = zlib.decompress(raw[20:])
    if len(blob) != n or zlib.crc32(blob) != crc:
        raise RuntimeError('%s: damaged track pack' % path)
    return blob
