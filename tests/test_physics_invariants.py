"""Solver sanity of the ODE-equivalent restatement (oracle/rb), which no reference artefact can pin because ODE itself
is absent from the reference tree ("solver parity unpinned", SURVEY.md section 8c: validation (i), (ii), (v))."""
import ctypes as C
import numpy as np
import pdb_ctypes as pc


def run(oracle, P, trk, s0, ticks, steer=0.0, a1=-1.0):
    h = oracle.cpuref_create(C.byref(P), trk, len(trk), C.byref(s0))
    for _ in range(ticks):
        oracle.cpuref_step_env(h, steer, a1)
    s = pc.DynState(); oracle.cpuref_get_state(h, C.byref(s))
    cs = pc.CarState(); oracle.cpuref_get_car_state(h, C.byref(cs))
    oracle.cpuref_destroy(h)
    return s, cs


def test_static_equilibrium_wheel_loads_carry_the_weight(oracle, env_params, flat_track, state0):
    """(i) settle on the flat plane at idle throttle: sum of tyre loads = total weight within 1e-3"""
    P = pc.CarParams.from_buffer_copy(bytes(env_params))
    s, cs = run(oracle, P, flat_track, state0, 1200, 0.0, -1.0)
    total = sum(P.bodies[b].mass for b in range(P.numBodies))
    g = abs(P.gravity[1])
    loads = np.array([s.tyre[i].load for i in range(4)])
    assert abs(loads.sum() - total * g) / (total * g) < 5e-3, (loads, total * g)
    # left/right: equal up to the drive-torque reaction of the live axle (env gas floor 0.1 keeps the car creeping)
    assert abs(loads[0] - loads[1]) / loads[0] < 0.10 and abs(loads[2] - loads[3]) / loads[2] < 0.35
    assert 0.15 < s.body[0].pos[1] < 1.0


def test_constraints_hold(oracle, env_params, flat_track, state0):
    """(ii) joint violations stay ERP-bounded while driving hard: DBall lengths and ball anchors"""
    P = pc.CarParams.from_buffer_copy(bytes(env_params))
    s, cs = run(oracle, P, flat_track, state0, 1500, 0.3, 1.0)

    def world(b, a):
        R = np.array(s.body[b].R[:]).reshape(3, 3)
        return np.array(s.body[b].pos[:]) + R @ np.array(a[:])
    worst = 0.0
    for j in range(P.numJoints):
        J = P.joints[j]
        if J.steerWheel >= 0:
            continue                                   # tie rods: anchor moves with the steering input
        if J.type == 3:                                # dball: |p2 - p1| = distance
            d = np.linalg.norm(world(J.b1, J.anchor2) - world(J.b0, J.anchor1))
            worst = max(worst, abs(d - J.distance))
        elif J.type == 1:                              # ball: anchors coincide
            worst = max(worst, np.linalg.norm(world(J.b1, J.anchor2) - world(J.b0, J.anchor1)))
    assert worst < 5e-3, worst
    for b in range(P.numBodies):
        R = np.array(s.body[b].R[:]).reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
        assert abs(np.linalg.norm(s.body[b].q[:]) - 1.0) < 1e-5


def test_straight_line_launch_is_straight_and_accelerates(oracle, env_params, flat_track, state0):
    """(v) zero steer, full throttle: stays on the centre line, gains speed, shifts up, no flags"""
    s, cs = run(oracle, env_params, flat_track, state0, 2000, 0.0, 1.0)
    assert abs(s.body[0].pos[0]) < 5.0          # wheelspin + axle torque reaction pull an uncorrected RWD launch sideways, but not off the 12 m track
    assert s.speed > 20.0 and s.currentGear >= 3
    assert cs.collisionFlag == 0 and cs.outOfTrackFlag == 0
    assert cs.engineRPM > 1000.0


import pytest
from conftest import car_params


@pytest.mark.parametrize('model', ['ks_toyota_ae86_drift', 'ks_mazda_rx7_tuned', 'gravygarage_street_ae86_readie'])
def test_free_flight_conserves_momentum_and_holds_the_joints(oracle, hostlib, flat_track, model):
    """(iii) bodies + joints only, gravity off, launched as one rigid tumbling motion: constraint forces are internal, so the
    linear momentum must be conserved to rounding, the angular momentum and the energy to the integrator's first-order error,
    and every joint must stay closed -- for the three joint topologies (33 / 26 / 38 rows)"""
    import ctypes as C
    P = car_params(model)
    s0 = pc.DynState(); assert hostlib.pdb_initial_state(C.byref(P), flat_track, C.byref(s0)) == 0
    h = oracle.cpuref_create(C.byref(P), flat_track, len(flat_track), C.byref(s0))
    v = np.array([10.0, 2.0, 5.0], np.float32); w = np.array([0.5, 1.0, -0.3], np.float32)
    out = np.zeros(16)
    assert oracle.cpuref_solver_freeflight(h, 1000, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    oracle.cpuref_destroy(h)
    p0, p1, l0, l1, e0, e1 = out[0:3], out[3:6], out[6:9], out[9:12], out[12], out[13]
    assert np.linalg.norm(p1 - p0) / np.linalg.norm(p0) < 1e-5, (p0, p1)
    assert np.linalg.norm(l1 - l0) / np.linalg.norm(l0) < 5e-5, (l0, l1)
    assert abs(e1 - e0) / e0 < 2e-4, (e0, e1)
    assert out[14] < 2e-3 and out[15] < 2e-3, (out[14], out[15])


@pytest.mark.parametrize('model', ['ks_toyota_ae86_drift', 'ks_mazda_rx7_tuned', 'gravygarage_street_ae86_readie'])
def test_fp32_factor_solve_agrees_with_fp64(oracle, hostlib, touge_track, model):
    """(iv) the canonical fp32 LDL^T (explicit fmaf, no pivoting) against a float64 solve of the very same system, sampled along a
    drive over the mountain road: the matrix is symmetric positive definite and the constraint impulses agree to 1e-4 of their
    scale -- the single-precision solve is not where accuracy is lost"""
    import ctypes as C
    P = car_params(model)
    s0 = pc.DynState(); assert hostlib.pdb_initial_state(C.byref(P), touge_track, C.byref(s0)) == 0
    h = oracle.cpuref_create(C.byref(P), touge_track, len(touge_track), C.byref(s0))
    o = pc.StepOut(); a = np.zeros(2, np.float32)
    worst, worst_cond = 0.0, 0.0
    for t in range(3000):
        oracle.cpuref_step_env(h, float(a[0]), float(a[1])); oracle.cpuref_get_out(h, C.byref(o))
        obs = np.array(o.obs[:], np.float32)
        oracle.cpuref_scenario_feedback(6, t, obs.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
        if t % 50 == 0:
            A = np.zeros((40, 40), np.float32).reshape(-1); rhs = np.zeros(40, np.float32); lam = np.zeros(40, np.float32)
            m = oracle.cpuref_last_system(h, A.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p), lam.ctypes.data_as(C.c_void_p), 40)
            assert m == P.numRows
            L = np.tril(A[:m * m].reshape(m, m).astype(np.float64))
            Af = L + np.tril(L, -1).T
            w = np.linalg.eigvalsh(Af)
            assert w[0] > 0.0                                   # SPD: the unpivoted factorisation is legitimate
            x = np.linalg.solve(Af, rhs[:m].astype(np.float64))
            worst = max(worst, float(np.abs(lam[:m] - x).max() / max(np.abs(x).max(), 1.0)))
            worst_cond = max(worst_cond, float(w[-1] / w[0]))
    oracle.cpuref_destroy(h)
    assert worst < 2e-4, (worst, worst_cond)     # measured: 9.5e-5 (AE86, condition number 1.2e6), 4e-6 (RX-7), 9e-6 (readie)


# ---- closed-form checks of single joints, independent of any car (and of pdrb's own bookkeeping): what a Ball, a Slider and a DBall
# must do by mechanics alone ----
def _joint_unit(oracle, kind, L=1.0, theta0=0.0, F=0.0, ticks=2000):
    out = (C.c_float * (4 * ticks))()
    assert oracle.cpuref_joint_unit(kind, C.c_float(L), C.c_float(theta0), C.c_float(F), ticks, out) == 0
    return np.array(out[:], dtype=np.float64).reshape(ticks, 4)


def test_ball_joint_pendulum_period(oracle):
    """a 1 kg cube (10 cm) on a 1 m massless rod through a Ball joint: physical-pendulum period 2 pi sqrt((I + m L^2) / (m g L)),
    amplitude correction (1 + theta0^2 / 16); the rod length holds to the ERP-bounded tolerance"""
    L, th = 1.0, 0.1
    r = _joint_unit(oracle, 0, L=L, theta0=th, ticks=3000)
    x = r[:, 0]
    up = np.nonzero((x[:-1] < 0) & (x[1:] >= 0))[0]             # upward zero crossings
    assert len(up) >= 4
    frac = up + (-x[up]) / (x[up + 1] - x[up])
    period = np.diff(frac).mean() / 333.0
    I = 1.0 / 12.0 * (0.1 ** 2 + 0.1 ** 2)
    expect = 2 * np.pi * np.sqrt((I + L * L) / (9.80665 * L)) * (1 + th * th / 16)
    assert abs(period - expect) / expect < 5e-3, (period, expect)
    assert np.abs(r[:, 2] - L).max() < 2e-3                     # the anchor holds


def test_slider_leaves_exactly_its_axis_free(oracle):
    """constant force along the slider axis: x = x0 + F t^2 / (2 m) (semi-implicit Euler: t (t + h) / 2), nothing moves across the
    axis, the bodies do not rotate against each other"""
    F, n = 3.0, 1500
    r = _joint_unit(oracle, 1, F=F, ticks=n)
    h = 1.0 / 333.0
    t = (np.arange(n) + 1) * h
    expect = 0.5 + 0.5 * F * t * (t + h)
    assert np.abs(r[:, 0] - expect).max() < 2e-3 * expect.max()
    assert np.abs(r[:, 1] - 0.2).max() < 1e-4 and np.abs(r[:, 2] + 0.1).max() < 1e-4
    assert r[:, 3].max() < 1e-5


def test_distance_joint_holds_its_length_under_load(oracle):
    """a DBall carrying 1 kg: the length drifts by no more than cfm * lambda * h / erp (~1e-8 m) plus float rounding"""
    r = _joint_unit(oracle, 2, L=0.7, ticks=2000)
    assert np.abs(r[:, 0] - 0.7).max() < 2e-6, np.abs(r[:, 0] - 0.7).max()
    assert np.abs(r[:, 1]).max() < 1e-6 and np.abs(r[:, 2]).max() < 1e-6
    assert np.abs(r[:, 3]).max() < 1e-6                         # the 1e9 kg anchor stays put
