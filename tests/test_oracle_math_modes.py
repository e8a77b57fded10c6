"""Links the two halves of the parity chain.  tests/test_oracle_golden.py pins the glibc build of the oracle to the
reference-TU trajectories bit for bit; tests/test_gpu_parity.py compares the GPU bit for bit with the portable-math
build (same restatement, elementary functions per csrc/device/pmath.hpp).  Here: the two oracle builds, stepped ONE
tick from the same state, differ by no more than the few 1-ulp libm disagreements can explain -- far below the 1e-4
bar -- at every tick of a driving scenario.  (Free-running they separate like any two roundings of a chaotic system:
that is a property of the vehicle, bounded below as well.)"""
import ctypes as C
import numpy as np
import oracle_ctypes
import pdb_ctypes as pc
import parity_util


def test_single_tick_deviation_between_math_modes(built, env_params, flat_track, state0):
    og = oracle_ctypes.load_oracle(False)
    op = oracle_ctypes.load_oracle(True)
    hg = og.cpuref_create(C.byref(env_params), flat_track, len(flat_track), C.byref(state0))
    hp = op.cpuref_create(C.byref(env_params), flat_track, len(flat_track), C.byref(state0))
    worst = 0.0
    rng = np.random.RandomState(5)
    steer, a1 = 0.0, 1.0
    s = pc.DynState()
    for t in range(900):
        if t % 150 == 0:
            steer, a1 = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.0, 1.0))
        og.cpuref_get_state(hg, C.byref(s))
        op.cpuref_set_state(hp, C.byref(s))            # re-synchronise: measure ONE tick of divergence
        og.cpuref_step_env(hg, steer, a1)
        op.cpuref_step_env(hp, steer, a1)
        sg, sp = pc.DynState(), pc.DynState()
        og.cpuref_get_state(hg, C.byref(sg)); op.cpuref_get_state(hp, C.byref(sp))
        rel, name, vg, vc, bad_int = parity_util.compare_states(sp, sg)
        assert not bad_int, (t, bad_int)
        worst = max(worst, rel)
    og.cpuref_destroy(hg); op.cpuref_destroy(hp)
    assert worst < 2e-5, worst


def test_free_running_math_modes_stay_physically_close(built, env_params, flat_track, state0):
    """3 s launch: both builds must tell the same story (speed, gear, distance) even though low bits differ"""
    og = oracle_ctypes.load_oracle(False)
    op = oracle_ctypes.load_oracle(True)
    hs = [lib.cpuref_create(C.byref(env_params), flat_track, len(flat_track), C.byref(state0)) for lib in (og, op)]
    for t in range(1000):
        og.cpuref_step_env(hs[0], 0.0, 1.0); op.cpuref_step_env(hs[1], 0.0, 1.0)
    a, b = pc.DynState(), pc.DynState()
    og.cpuref_get_state(hs[0], C.byref(a)); op.cpuref_get_state(hs[1], C.byref(b))
    og.cpuref_destroy(hs[0]); op.cpuref_destroy(hs[1])
    assert a.currentGear == b.currentGear
    assert a.speed > 5.0 and abs(a.speed - b.speed) / a.speed < 1e-2
    assert abs(a.body[0].pos[2] - b.body[0].pos[2]) < 0.05 * abs(a.body[0].pos[2] - state0.body[0].pos[2]) + 0.05
