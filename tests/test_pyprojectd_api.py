"""The PyProjectD-compatible module (csrc/pybind/pyprojectd.cpp): same names and error convention as the reference's
pybind11 module (reference src/PyProjectD/PyProjectD.cpp:515-640).  CPU part: surface + conventions; the GPU part
(-m gpu) drives the env classes and checks every observation/reward against the oracle."""
import ctypes as C, os, sys, tempfile
import numpy as np
import pytest

REFERENCE_FUNCTIONS = [   # m.def(...) names, reference PyProjectD.cpp:600-640
    'setSeed', 'setLogFile', 'clearLogFile', 'writeLog', 'createSimulator', 'destroySimulator', 'stepSimulator', 'loadTrack', 'unloadTrack',
    'addCar', 'removeCar', 'teleportCarToLocation', 'teleportCarToPits', 'teleportCarToSpline', 'teleportCarByMode', 'setCarAutoTeleport',
    'setCarControls', 'setCarAssists', 'getCarState', 'setCarRawTune', 'setCarTune', 'setScoringVar', 'getScoringVar',
    'launchPlaygroundInOwnThread', 'initPlayground', 'shutPlayground', 'shutAll', 'tickPlayground', 'isPlaygroundInitialized', 'isPlaygroundExited',
    'moveWindow', 'resizeWindow', 'setRenderHz', 'setActiveSimulator', 'setActiveCar', 'getActiveSimulator', 'getActiveCar']
CARSTATE_FIELDS = [       # def_readonly names, reference PyProjectD.cpp:556-598
    'carId', 'simId', 'timestamp', 'controls', 'collisionFlag', 'outOfTrackFlag', 'trackPointId', 'lastTrackPointTimestamp', 'trackLocation',
    'bodyVsTrack', 'velocityVsTrack', 'engineRPM', 'speedMS', 'gear', 'gearGrinding', 'bodyMatrix', 'bodyPos', 'bodyEuler', 'accG', 'velocity',
    'localVelocity', 'angularVelocity', 'localAngularVelocity', 'hubMatrix', 'tyreContacts', 'tyreLoad', 'tyreAngularSpeed', 'tyreSlipRatio',
    'tyreNdSlip', 'probes', 'lookAhead', 'stepReward', 'totalReward']


@pytest.fixture(scope='module')
def pd(built):
    import PyProjectD
    return PyProjectD


@pytest.fixture()
def base(built):
    import synthetic_tracks
    d = tempfile.mkdtemp(prefix='pdb_env_')
    synthetic_tracks.make_base(d, tracks=('flat',))
    synthetic_tracks.install_packed_car(d)
    return d


def test_module_surface(pd):
    for f in REFERENCE_FUNCTIONS:
        assert callable(getattr(pd, f)), f
    for c in ('vec3f', 'mat44f', 'CarControls', 'CarState'):
        assert hasattr(pd, c)
    s = pd.CarState()
    for f in CARSTATE_FIELDS:
        assert hasattr(s, f), f
    with pytest.raises(AttributeError):
        s.gear = 3                                   # read-only like the reference
    c = pd.CarControls()
    assert (c.steer, c.clutch, c.brake, c.handBrake, c.gas, c.isShifterSupported, c.requestedGearIndex, c.gearUp, c.gearDn) == (0, 0, 0, 0, 0, 1, -1, 0, 0)
    c.steer = 0.25; c.requestedGearIndex = 2
    assert c.steer == 0.25 and c.requestedGearIndex == 2
    m = pd.mat44f(); assert (m.M11, m.M22, m.M33, m.M44, m.M12) == (1, 1, 1, 1, 0)
    assert len(s.probes) == 10 and len(s.lookAhead) == 5 and len(s.hubMatrix) == 4 and len(s.tyreContacts) == 4


def test_error_convention(pd, base):
    """creators return -1 and log; everything else ignores unknown ids; nothing throws (PyProjectD.cpp:74-109,111-137,219-237)"""
    log = os.path.join(base, 'log.txt')
    pd.setLogFile(log, True)
    assert pd.createSimulator('/no/such/base') == -1
    pd.stepSimulator(12345, 1.0 / 333.0)
    pd.loadTrack(12345, 'flat')
    assert pd.addCar(12345, 'ks_toyota_ae86_drift') == -1
    st = pd.CarState(); pd.getCarState(12345, 0, st)
    assert pd.getScoringVar(12345, 0, 'TravelBonus') == 0.0
    pd.setCarTune(12345, 0, 'FRONT_BIAS', 55.0); pd.teleportCarByMode(12345, 0, 0); pd.destroySimulator(12345)
    sim = pd.createSimulator(base)
    assert sim >= 0
    assert pd.addCar(sim, 'ks_toyota_ae86_drift') == -1          # no track yet
    pd.loadTrack(sim, 'no_such_track')
    assert pd.addCar(sim, 'ks_toyota_ae86_drift') == -1
    pd.loadTrack(sim, 'flat')
    assert pd.addCar(sim, 'no_such_car') == -1
    car = pd.addCar(sim, 'ks_toyota_ae86_drift')
    assert car == 0
    assert pd.addCar(sim, 'ks_toyota_ae86_drift') == 1           # a second car of the simulator (cfg/sim.ini MAX_CARS = 2) ...
    assert pd.addCar(sim, 'ks_toyota_ae86_drift') == -1          # ... and the simulator is full
    pd.removeCar(sim, 1)
    assert pd.addCar(sim, 'no_such_car') == -1
    assert pd.addCar(sim, 'ks_toyota_ae86_drift') == 1
    pd.removeCar(sim, 1)
    pd.setScoringVar(sim, car, 'TravelBonus', 0.25)
    assert pd.getScoringVar(sim, car, 'TravelBonus') == 0.25
    assert pd.getScoringVar(sim, car, 'NoSuchVar') == 0.0
    pd.setCarTune(sim, car, 'FRONT_BIAS', 55.0)                   # packed car: refused + logged, no exception
    pd.writeLog('[ENV] hello')
    sim2 = pd.createSimulator(base)
    assert sim2 == sim + 1                                        # monotonically increasing ids
    pd.destroySimulator(sim); pd.destroySimulator(sim2)
    text = open(log).read()
    assert 'EXCEPTION' in text and '[ENV] hello' in text and 'createSimulator' in text
    pd.setLogFile('', False)


def test_no_cpu_fallback_through_the_module(pd, base):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    sim = pd.createSimulator(base); pd.loadTrack(sim, 'flat'); car = pd.addCar(sim, 'ks_toyota_ae86_drift')
    st0 = pd.CarState(); pd.getCarState(sim, car, st0)
    pd.stepSimulator(sim, 1.0 / 333.0)                           # logs the failure, does not compute anything
    st = pd.CarState(); pd.getCarState(sim, car, st)
    assert st.timestamp == st0.timestamp == 0.0
    assert pd.createBatch(sim, 4, 0) == -1
    pd.destroySimulator(sim)
    import projectd_env
    with pytest.raises(RuntimeError):
        projectd_env.ProjectDVecEnv(4, base, track_name='flat')


def test_env_config_and_bounds(built):
    import projectd_env as E
    cfg = E.EnvConfig(track_name='flat', terminate_when_stuck=False)
    assert cfg.track_name == 'flat' and cfg.car_model == 'ks_toyota_ae86_drift' and cfg.stuck_timeout == 5.0
    assert cfg.scoring_vars['TravelBonus'] == 0.1 and cfg.car_tunes['ks_toyota_ae86_drift']['FINAL_RATIO'] == 5.0
    with pytest.raises(TypeError):
        E.EnvConfig(no_such_setting=1)
    lo, hi = E.obs_bounds(cfg)
    assert lo.shape == hi.shape == (24,) and hi[0] == 100 and hi[6] == 10 and lo[6] == 0 and abs(hi[12] - np.pi) < 1e-6 and hi[17] == 50 and lo[17] == 0
    assert E.linscale(0.0, -1.0, 1.0, 0.1, 1.0) == 0.55 and E.linscale(-5.0, -1.0, 1.0, 0.1, 1.0) == 0.1


@pytest.mark.gpu
def test_single_env_matches_oracle(pd, base):
    """ProjectDEnv through the classic calls: obs / reward of every step equal the oracle's (bit-exact)"""
    import projectd_env as E, pdbatch, pdb_ctypes as pc, oracle_ctypes
    env = E.ProjectDEnv(base, track_name='flat')
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    h = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0))
    o = pc.StepOut()
    obs = env.reset()
    orc.cpuref_step_env(h, 0.0, 0.0); orc.cpuref_get_out(h, C.byref(o))
    assert np.array_equal(obs, np.array(o.obs[:], np.float32))
    rng = np.random.RandomState(3)
    a = np.array([0.0, 1.0], np.float32)
    for t in range(400):
        if t % 60 == 0:
            a = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-1, 1)], np.float32)
        obs, rew, term, trunc, info = env.step(a)
        orc.cpuref_step_env(h, float(a[0]), float(a[1])); orc.cpuref_get_out(h, C.byref(o))
        assert np.array_equal(obs, np.array(o.obs[:], np.float32)), t
        if not term:
            assert np.float32(rew) == np.float32(o.reward), t
        assert env.dstate.gear >= 0 and env.dstate.simId == env.sim
    orc.cpuref_destroy(h)
    env.close()


@pytest.mark.gpu
def test_teleports_to_pits_and_locations_through_the_module(pd, built):
    """teleportCarToPits / teleportCarToLocation (reference PyProjectD.cpp:250-266 -> Car::teleportToPits / Car::forcePosition, Car.cpp:1240-1323) through the
    module on the mountain road, whose pits.ini holds five boxes: the CarState of every tick equals the oracle's, the oracle's record going through the host
    library's pdb_teleport_to_pit / pdb_teleport_to_location (themselves pinned to the reference TUs by the `pits*` / `locations*` goldens).  An id
    outside the list moves nothing, like the reference."""
    import synthetic_tracks, pdbatch, pdb_ctypes as pc, oracle_ctypes
    base = tempfile.mkdtemp(prefix='pdb_pits_')
    synthetic_tracks.make_base(base, tracks=('touge',)); synthetic_tracks.install_packed_car(base)
    sim = pd.createSimulator(base); pd.loadTrack(sim, 'touge'); car = pd.addCar(sim, 'ks_toyota_ae86_drift')
    assert sim >= 0 and car == 0
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
    P = pdbatch.packed_params(); trk = pc.build_track(lib, base, 'touge')
    assert lib.pdb_track_num_pits(trk) == 5
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    h = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0))
    ctl = pd.CarControls(); ctl.gas = 0.8; ctl.steer = 0.1
    a8 = np.array([0.1, 0, 0, 0, 0.8, -1, 0, 0], np.float32)
    st = pd.CarState(); cs = pc.CarState(); s = pc.DynState()

    def run(n):
        for _ in range(n):
            pd.setCarControls(sim, car, True, ctl); pd.stepSimulator(sim, 1.0 / 333.0); pd.getCarState(sim, car, st)
            orc.cpuref_step_controls(h, a8.ctypes.data_as(C.c_void_p)); orc.cpuref_get_car_state(h, C.byref(cs))
            for f in ('bodyPos', 'velocity', 'bodyEuler', 'localAngularVelocity'):
                v = getattr(st, f)
                assert (np.float32(v.x), np.float32(v.y), np.float32(v.z)) == tuple(np.float32(x) for x in getattr(cs, f)), f
            assert np.float32(st.engineRPM) == np.float32(cs.engineRPM) and st.gear == cs.gear and np.float32(st.trackLocation) == np.float32(cs.trackLocation)
    run(200)
    before = (st.bodyPos.x, st.bodyPos.y, st.bodyPos.z)
    pd.teleportCarToPits(sim, car, 99); pd.teleportCarToPits(sim, car, -1)      # outside the list: nothing happens
    run(1)
    assert abs(st.bodyPos.x - before[0]) < 0.1 and abs(st.bodyPos.z - before[2]) < 0.1
    for pit in (1, 4, 0):
        m = (C.c_float * 16)(); assert lib.pdb_track_pit(trk, pit, m) == 0
        pd.teleportCarToPits(sim, car, pit)
        orc.cpuref_get_state(h, C.byref(s)); assert lib.pdb_teleport_to_pit(C.byref(P), trk, pit, C.byref(s)) == 0; orc.cpuref_set_state(h, C.byref(s))
        run(1)
        assert abs(st.bodyPos.x - m[12]) < 0.05 and abs(st.bodyPos.z - m[14]) < 0.05 and st.speedMS < 0.5   # the car stands in the box
        run(150)
    for off in ((3.0, 4.0, -2.0), (-1.0, 0.2, 6.0)):
        x, y, z = (float(np.float32(st.bodyPos.x) + np.float32(off[0])), float(np.float32(st.bodyPos.y) + np.float32(off[1])), float(np.float32(st.bodyPos.z) + np.float32(off[2])))
        pd.teleportCarToLocation(sim, car, x, y, z)
        orc.cpuref_get_state(h, C.byref(s)); assert lib.pdb_teleport_to_location(C.byref(P), trk, C.c_float(x), C.c_float(y), C.c_float(z), C.byref(s)) == 0
        orc.cpuref_set_state(h, C.byref(s))
        run(1)
        assert abs(st.bodyPos.x - x) < 0.05 and abs(st.bodyPos.z - z) < 0.05
        run(150)
    orc.cpuref_destroy(h); pd.destroySimulator(sim)


@pytest.mark.gpu
def test_two_cars_in_one_simulator_through_the_module(pd, built):
    """Simulator::addCar twice (PyProjectD.cpp:219-237; cfg/sim.ini MAX_CARS = 2): the second car is put down 10 m ahead on the plane, the first closes in flat out
    through its wake -- Car::updateAirPressure / Sim/SlipStream.cpp, pinned to the reference TUs by the twocar_* goldens -- and every CarState of both cars equals two
    oracle cars that exchange their wakes every tick; the air really is thinner for the car behind (its drag drops: it ends up faster than it would alone)."""
    import synthetic_tracks, pdbatch, pdb_ctypes as pc, oracle_ctypes
    base = tempfile.mkdtemp(prefix='pdb_two_')
    synthetic_tracks.make_base(base, tracks=('flat',)); synthetic_tracks.install_packed_car(base)
    sim = pd.createSimulator(base); pd.loadTrack(sim, 'flat')
    assert pd.addCar(sim, 'ks_toyota_ae86_drift') == 0 and pd.addCar(sim, 'ks_toyota_ae86_drift') == 1
    pd.teleportCarToSpline(sim, 1, 0.0034)
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
    P = pdbatch.packed_params(); trk = pc.build_track(lib, base, 'flat')
    S = [pc.DynState(), pc.DynState()]
    for c in range(2):
        assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S[c])) == 0
    assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(0.0034), C.byref(S[1])) == 0
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S[c])) for c in range(2)]
    orc.cpuref_set_guid(hs[1], 1)
    solo = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S[0]))   # the first car alone, same inputs: no wake to run in
    ctl = [pd.CarControls(), pd.CarControls()]
    ctl[0].gas = 1.0; ctl[1].gas = 0.3
    a8 = [np.array([0, 0, 0, 0, 1.0, -1, 0, 0], np.float32), np.array([0, 0, 0, 0, 0.3, -1, 0, 0], np.float32)]
    st = pd.CarState(); cs = pc.CarState(); slips = (pc.SlipState * 2)()
    for t in range(3600):
        for c in range(2):
            pd.setCarControls(sim, c, True, ctl[c])
        pd.stepSimulator(sim, 1.0 / 333.0)
        for c in range(2):
            orc.cpuref_get_slip(hs[c], C.byref(slips[c]))
        for c in range(2):
            orc.cpuref_set_other_slips(hs[c], C.byref(slips[1 - c]), 1)
        for c in range(2):
            orc.cpuref_step_controls(hs[c], a8[c].ctypes.data_as(C.c_void_p))
        orc.cpuref_step_controls(solo, a8[0].ctypes.data_as(C.c_void_p))
        if t % 5 == 0 or t > 3590:
            for c in range(2):
                pd.getCarState(sim, c, st); orc.cpuref_get_car_state(hs[c], C.byref(cs))
                assert st.carId == c
                for f in ('bodyPos', 'velocity', 'localAngularVelocity'):
                    v = getattr(st, f)
                    assert (np.float32(v.x), np.float32(v.y), np.float32(v.z)) == tuple(np.float32(x) for x in getattr(cs, f)), (t, c, f)
                assert np.float32(st.engineRPM) == np.float32(cs.engineRPM) and st.gear == cs.gear, (t, c)
    pd.getCarState(sim, 0, st); orc.cpuref_get_car_state(solo, C.byref(cs))
    assert st.speedMS > cs.speedMS, (st.speedMS, cs.speedMS)   # the tow: a little faster than the same car with the same inputs alone
    for h in hs + [solo]:
        orc.cpuref_destroy(h)
    pd.destroySimulator(sim)


@pytest.mark.gpu
def test_a_lane_takes_a_configured_simulators_whole_setup(pd, base):
    """setBatchLaneSetup (+ setBatchLaneTune): a lane of simulator A's batch steps exactly like a batch made from simulator B, whose springs, bar, dampers, gear ratio
    and brake power were tuned with the reference's own setCarRawTune; the other lanes are untouched"""
    simA = pd.createSimulator(base); pd.loadTrack(simA, 'flat'); carA = pd.addCar(simA, 'ks_toyota_ae86_drift')
    simB = pd.createSimulator(base); pd.loadTrack(simB, 'flat'); carB = pd.addCar(simB, 'ks_toyota_ae86_drift')
    for name, v in (('SPRING_RATE_LF', 52000.0), ('SPRING_RATE_RF', 47000.0), ('ARB_FRONT', 9000.0), ('DAMP_BUMP_LR', 2100.0), ('DAMP_REBOUND_RR', 5200.0),
                    ('INTERNAL_GEAR_2', 3.1), ('BRAKE_POWER_MULT', 0.8), ('ROD_LENGTH_RF', 0.004), ('CAMBER_LF', -0.05), ('FINAL_RATIO', 4.6)):
        pd.setCarRawTune(simB, carB, name, v)
    n = 4
    bA = pd.createBatch(simA, n, 0); bB = pd.createBatch(simB, n, 0); bP = pd.createBatch(simA, n, 0)
    assert min(bA, bB, bP) >= 0
    assert pd.setBatchLaneTune(bA, 2, simB) and pd.setBatchLaneSetup(bA, 2, simB)
    acts = np.tile(np.array([[0.12, 0.4]], np.float32), (n, 1))
    differs = False
    for t in range(400):
        oA = pd.stepBatch(bA, acts); oB = pd.stepBatch(bB, acts); oP = pd.stepBatch(bP, acts)
        assert np.array_equal(oA[2].view(np.int32), oB[0].view(np.int32)), t
        assert np.array_equal(oA[0].view(np.int32), oP[0].view(np.int32)) and np.array_equal(oA[3].view(np.int32), oP[3].view(np.int32)), t
        differs = differs or not np.array_equal(oA[2], oA[0])
    assert differs
    assert pd.setBatchLaneSetup(bA, 2, -1) and not pd.setBatchLaneSetup(bA, 9, simB)
    for b in (bA, bB, bP):
        pd.destroyBatch(b)
    pd.destroySimulator(simA); pd.destroySimulator(simB)


@pytest.mark.gpu
def test_vec_env_matches_oracle_and_resets(pd, base):
    import projectd_env as E, pdbatch, pdb_ctypes as pc, oracle_ctypes, sharding
    n = 16
    env = E.ProjectDVecEnv(n, base, track_name='flat', terminate_low_reward=-1e9)
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n)]
    o = pc.StepOut()
    obs = env.reset()
    for i in range(n):
        orc.cpuref_step_env(hs[i], 0.0, 0.0); orc.cpuref_get_out(hs[i], C.byref(o))
        assert np.array_equal(obs[i], np.array(o.obs[:], np.float32))
    acts = sharding.global_actions(n, 77)
    for t in range(300):
        obs, rew, term, trunc, info = env.step(acts)
        assert obs.shape == (n, 24) and rew.shape == (n,) and term.dtype == bool
        for i in range(n):
            orc.cpuref_step_env(hs[i], float(acts[i, 0]), float(acts[i, 1])); orc.cpuref_get_out(hs[i], C.byref(o))
            if not term[i]:
                assert np.array_equal(obs[i], np.array(o.obs[:], np.float32)), (t, i)
                assert rew[i] == np.float32(np.float64(o.reward)), (t, i)
        if term.any():
            break
    # masked reset: only the selected lanes return to the start line; they take their zero-action tick in the next step
    m = np.zeros(n, np.uint8); m[[1, 5]] = 1
    cs = pd.CarState()
    pd.getBatchCarState(env.batch, 0, cs); z0 = cs.bodyPos.z
    assert env.reset(m) is None
    pd.getBatchCarState(env.batch, 1, cs); assert abs(cs.bodyPos.z + 1500.0) < 1.0
    pd.getBatchCarState(env.batch, 0, cs); assert cs.bodyPos.z == z0
    # ... and go on from the pose the caller gave them (the kernel's own re-creation of a faulted record must not take a caller's reset tick for
    # its own: ADVICE r5): the masked lanes' next tick is the reset tick -- zero action, reward 0 -- on the teleported record, everything
    # Car::reset keeps (tyre temperatures, the steering filter, ...) kept; then every lane steps on, observation for observation like the oracle
    assert env.kernel_env
    for i in (1, 5):
        s = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(s))
        assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(0.0), C.byref(s)) == 0
        orc.cpuref_set_state(hs[i], C.byref(s))
    for t in range(60):
        obs, rew, term, trunc, info = env.step(acts)
        for i in range(n):
            zero = t == 0 and i in (1, 5)
            orc.cpuref_step_env(hs[i], 0.0 if zero else float(acts[i, 0]), 0.0 if zero else float(acts[i, 1])); orc.cpuref_get_out(hs[i], C.byref(o))
            if not term[i] or zero:
                assert np.array_equal(obs[i], np.array(o.obs[:], np.float32)), (t, i)
            if zero:
                assert rew[i] == 0.0 and not term[i]
    for h in hs:
        orc.cpuref_destroy(h)
    env.close()


@pytest.mark.gpu
def test_vec_env_episodes_with_auto_reset_match_oracle(pd, base):
    """4000 ticks of 24 lanes on the mountain road under a deliberately poor policy (the feedback controller plus steering
    noise), so that lanes leave the road, terminate and are reset at different times: every observation, reward and
    termination of ProjectDVecEnv equals an oracle-side replay of the reference env logic (projectd_env.py:157-227)."""
    import projectd_env as E, pdbatch, pdb_ctypes as pc, oracle_ctypes, synthetic_tracks
    synthetic_tracks.make_base(base, tracks=('flat', 'touge'))
    n = 24
    env = E.ProjectDVecEnv(n, base, track_name='touge')
    P = pdbatch.packed_params(); trk = pc.build_track(pc.load_product(host_only=True), base, 'touge')
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n)]
    o = pc.StepOut()
    obs = env.reset()
    ref_obs = np.zeros((n, 24), np.float32)
    for i in range(n):
        orc.cpuref_step_env(hs[i], 0.0, 0.0); orc.cpuref_get_out(hs[i], C.byref(o)); ref_obs[i] = o.obs[:]
    assert np.array_equal(obs, ref_obs)
    total = np.zeros(n); pending = np.zeros(n, bool)
    rng = np.random.RandomState(9)
    noise = np.zeros(n, np.float32); episodes = 0
    a = np.zeros((n, 2), np.float32)
    for t in range(4000):
        if t % 50 == 0:
            noise = (rng.uniform(-1, 1, n) * (rng.uniform(0, 1, n) < 0.3) * 0.6).astype(np.float32)
        for i in range(n):
            orc.cpuref_scenario_feedback(6, t, obs[i].ctypes.data_as(C.c_void_p), a[i].ctypes.data_as(C.c_void_p))
        a[:, 0] = np.clip(a[:, 0] + noise, -1, 1)
        a[:8, 0] = np.where(np.arange(8) % 2 == 0, 0.8, -0.8); a[:8, 1] = 1.0      # eight lanes drive straight off the road, again and again
        obs, rew, term, trunc, info = env.step(a)
        for i in range(n):
            ai = (0.0, 0.0) if pending[i] else (float(a[i, 0]), float(a[i, 1]))
            orc.cpuref_step_env(hs[i], ai[0], ai[1]); orc.cpuref_get_out(hs[i], C.byref(o))
            assert np.array_equal(obs[i], np.array(o.obs[:], np.float32)), (t, i)
            r = float(np.float32(o.reward)); done = False
            if o.flags & 1: r -= 50.0; done = True
            if o.flags & 2: r -= 50.0; done = True
            if o.flags & 4: r -= 50.0; done = True
            total[i] += r
            if total[i] < -200.0: done = True
            if pending[i]:
                r = 0.0; done = False; total[i] = 0.0; pending[i] = False
            assert rew[i] == np.float32(r) and bool(term[i]) == done, (t, i, rew[i], r, term[i], done)
            if done:   # reference reset: teleportCarByMode(Start) now, the zero-action tick on the next step
                s = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(s))
                assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(0.0), C.byref(s)) == 0
                orc.cpuref_set_state(hs[i], C.byref(s)); pending[i] = True; episodes += 1
    assert episodes >= 5, episodes          # the noisy lanes really did crash out and restart
    for h in hs:
        orc.cpuref_destroy(h)
    env.close()


@pytest.mark.gpu
def test_torch_env_equals_numpy_env(pd, base):
    """the device-resident env (zero-copy tensors, bookkeeping in torch ops) and ProjectDVecEnv (host arrays through the
    PyProjectD-compatible module) produce the same observations, rewards and terminations, episode ends and resets included"""
    import torch, projectd_env as E, projectd_torch_env as TE, pdbatch, pdb_ctypes as pc, synthetic_tracks
    synthetic_tracks.make_base(base, tracks=('flat', 'touge'))
    n = 24
    envA = E.ProjectDVecEnv(n, base, track_name='touge')
    P = pdbatch.packed_params(); trk = pc.build_track(pc.load_product(host_only=True), base, 'touge')
    envB = TE.ProjectDTorchVecEnv(n, P, trk, device=0)
    obsA = envA.reset(); obsB = envB.reset()
    assert np.array_equal(obsA, obsB.cpu().numpy())
    a = np.zeros((n, 2), np.float32); ends = 0
    for t in range(2500):
        a[:, 0] = np.where(np.arange(n) % 3 == 0, 0.7, 0.02 * np.sin(0.01 * t + np.arange(n))); a[:, 1] = np.where(np.arange(n) % 3 == 0, 1.0, 0.2)
        oA, rA, tA, _, _ = envA.step(a)
        oB, rB, tB, _ = envB.step(torch.from_numpy(a).to('cuda:0'))
        assert np.array_equal(oA, oB.cpu().numpy()), t
        assert np.array_equal(rA, rB.cpu().numpy()) and np.array_equal(tA, tB.cpu().numpy()), t
        ends += int(tA.sum())
    assert ends >= 3
    envA.close(); envB.close()


@pytest.mark.gpu
def test_vec_env_terminates_on_hit(pd, base):
    """walled strip, env defaults (terminate_on_hit, projectd_env.py:38,186-189): lanes end their episodes on the ridge (belly
    box) or at a wall (hull) with the hit penalty, are teleported back and start over -- every observation, reward and
    termination equal to an oracle-side replay, and collisions are what ended the episodes"""
    import projectd_env as E, pdbatch, pdb_ctypes as pc, oracle_ctypes, synthetic_tracks
    synthetic_tracks.make_base(base, tracks=('flat', 'walled'))
    n = 12
    env = E.ProjectDVecEnv(n, base, track_name='walled', terminate_off_track=False)
    P = pdbatch.packed_params(); trk = pc.build_track(pc.load_product(host_only=True), base, 'walled')
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n)]
    o = pc.StepOut()
    obs = env.reset()
    for i in range(n):
        orc.cpuref_step_env(hs[i], 0.0, 0.0)
    total = np.zeros(n); pending = np.zeros(n, bool); hits = 0; other = 0
    a = np.zeros((n, 2), np.float32)
    a[:, 0] = np.linspace(-0.25, 0.25, n); a[:, 1] = 1.0
    for t in range(2200):
        obs, rew, term, trunc, info = env.step(a)
        for i in range(n):
            ai = (0.0, 0.0) if pending[i] else (float(a[i, 0]), float(a[i, 1]))
            orc.cpuref_step_env(hs[i], ai[0], ai[1]); orc.cpuref_get_out(hs[i], C.byref(o))
            assert np.array_equal(obs[i], np.array(o.obs[:], np.float32)), (t, i)
            r = float(np.float32(o.reward)); done = False
            if o.flags & 1: r -= 50.0; done = True
            if o.flags & 4: r -= 50.0; done = True
            total[i] += r
            if total[i] < -200.0: done = True
            if pending[i]:
                r = 0.0; done = False; total[i] = 0.0; pending[i] = False
            assert rew[i] == np.float32(r) and bool(term[i]) == done, (t, i, rew[i], r, term[i], done)
            if done:
                hits += int(o.flags & 1 != 0); other += int(o.flags & 1 == 0)
                s = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(s))
                assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(0.0), C.byref(s)) == 0
                assert all(s.damageZoneLevel[k] == 0 for k in range(5))      # Car::reset clears the damage (Car.cpp:403-407)
                orc.cpuref_set_state(hs[i], C.byref(s)); pending[i] = True
    assert hits >= n, (hits, other)          # every lane reaches the ridge or a wall at least once
    for h in hs:
        orc.cpuref_destroy(h)
    env.close()
