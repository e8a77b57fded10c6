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
