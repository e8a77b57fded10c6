"""Collision response: the contact joints PhysicsEngineODE::onCollision creates
(Physics/ODE/PhysicsEngineODE.cpp:283-331) and the bounded rows they add to dWorldStep.  ODE is not in the reference tree, so
nothing here can be compared with it: the rows follow ODE's published contact joint (joints/contact.cpp), the LCP is solved by
this project's method (oracle/rb/pdrb.cpp solveContacts).  What is checked: (1) the rows against closed
forms on a lone body, (2) the LCP solution against an independent float64 active-set solve of the same matrices, (3) the
invariants a contact solve owes -- lambda_n >= 0, |lambda_t| <= mu lambda_n(no friction), no penetration growth, no energy
from nothing, the car stays on its side of the wall, a hit does not create kinetic energy beyond the capped push-out -- along drives of the CPU oracle on the walled strip."""
import ctypes as C, os
import numpy as np
import pytest
import pdb_ctypes as pc
from conftest import car_params

AE86 = 'ks_toyota_ae86_drift'
H = np.float32(1.0 / 333.0)
FPS = np.float32(1.0) / H


def _unit(oracle, mass, sides, pos, lvel, avel, contacts, gravity=True):
    n = len(contacts)
    cs = (pc.Contact * max(n, 1))()
    for i, (p, nrm, depth, kind) in enumerate(contacts):
        cs[i].pos[:] = p; cs[i].normal[:] = nrm; cs[i].depth = depth; cs[i].kind = kind
    st = (C.c_float * 9)(*pos, *lvel, *avel)
    out = (C.c_float * (9 + 9 * max(n, 1)))()
    it = oracle.cpuref_contact_unit(C.c_float(mass), (C.c_float * 3)(*sides), st, cs, n, C.c_float(float(H)), 1 if gravity else 0, out)
    o = np.array(out[:], dtype=np.float64)
    return dict(lvel=o[0:3], avel=o[3:6], pos=o[6:9], lam=o[9:9 + 3 * n], lo=o[9 + 3 * n:9 + 6 * n], hi=o[9 + 6 * n:9 + 9 * n], it=it)


def test_normal_row_closed_form(oracle):
    """one contact under the centre of mass, no tangential motion: lambda = (c/h - (v/h + g)) / (1/m + cfm/h) with
    c = max(min(fps * erp * depth, 3), bounce * approach speed); both surface kinds of PhysicsEngineODE.cpp:295-322"""
    m, g = 1000.0, -9.80665
    for kind, erp, cfm, bounce in ((0, 0.3, 1.0e-4, 0.01), (1, 0.714285731, 0.000952380942, 0.0)):
        for depth, vy in ((0.01, -2.0), (0.2, -0.5), (0.0, -1.0), (0.002, 0.3)):
            r = _unit(oracle, m, (1.4, 1.35, 4.7), (0, 1, 0), (0, vy, 0), (0, 0, 0), [((0, 0.3, 0), (0, 1, 0), depth, kind)])
            c = min(float(FPS) * erp * depth, 3.0)
            if -vy > 0:
                c = max(c, bounce * -vy)
            lam = (c * float(FPS) - (vy * float(FPS) + g)) / (1.0 / m + cfm * float(FPS))
            lam = max(lam, 0.0)
            assert r['lam'][0] == pytest.approx(lam, rel=2e-5, abs=1e-2), (kind, depth, vy)
            assert r['lvel'][1] == pytest.approx(vy + float(H) * (g + lam / m), rel=1e-5, abs=1e-6)
            assert abs(r['lam'][1]) < 1e-3 and abs(r['lam'][2]) < 1e-3      # nothing to resist sideways


def test_separating_contact_carries_no_force(oracle):
    r = _unit(oracle, 1000.0, (1.4, 1.35, 4.7), (0, 1, 0), (0, 5.0, 0), (0, 0, 0), [((0, 0.3, 0), (0, 1, 0), 0.001, 0)])
    assert r['lam'][0] == 0.0 and r['lvel'][1] == pytest.approx(5.0 + float(H) * -9.80665, rel=1e-6)


def test_friction_limits_follow_the_frictionless_normal_force(oracle):
    """dContactApprox1: the friction rows are boxed by mu * |lambda_n| of the solve WITHOUT friction; sliding saturates them
    against the motion, a slow creep is stopped inside the box"""
    m = 1000.0
    for kind, mu in ((0, 0.25), (1, 0.1)):
        r0 = _unit(oracle, m, (1.4, 1.35, 4.7), (0, 1, 0), (0, -1.0, 0), (0, 0, 0), [((0, 0.3, 0), (0, 1, 0), 0.01, kind)])
        lam_n0 = r0['lam'][0]
        r = _unit(oracle, m, (1.4, 1.35, 4.7), (0, 1, 0), (8.0, -1.0, 0), (0, 0, 0), [((0, 0.3, 0), (0, 1, 0), 0.01, kind)])
        assert r['hi'][1] == pytest.approx(mu * lam_n0, rel=1e-5) and r['lo'][1] == pytest.approx(-mu * lam_n0, rel=1e-5)
        # dPlaneSpace((0,1,0)) = t1 (-1,0,0), t2 (0,0,1): sliding along +x saturates row 1 at +hi (force along t1 = -x)
        assert r['lam'][1] == pytest.approx(r['hi'][1], rel=1e-6)
        assert r['lvel'][0] < 8.0
        creep = _unit(oracle, m, (1.4, 1.35, 4.7), (0, 1, 0), (0.001, -1.0, 0), (0, 0, 0), [((0, 1.0, 0), (0, 1, 0), 0.01, kind)])
        assert abs(creep['lam'][1]) < creep['hi'][1] and abs(creep['lvel'][0]) < 1e-4   # held: inside the box


def test_lcp_solution_against_float64_active_set(oracle):
    """many contacts at once on a tumbling body: the fp32 solver's result satisfies the box LCP's conditions for the
    float64 rebuild of the same rows (normal stage and friction stage), within fp32 accuracy"""
    rng = np.random.RandomState(7)
    worst_iter = 0
    for trial in range(60):
        # up to the cap of 32 contacts (96 rows); the later trials repeat contact points exactly, as two wall triangles sharing a pierced edge do --
        # rows in exact pairs, the case block pivoting cycled on
        n = int(rng.randint(1, 11)) if trial < 30 else int(rng.randint(11, 33))
        contacts = []
        for i in range(n):
            if trial >= 45 and i >= 2 and rng.uniform() < 0.5:
                contacts.append(contacts[int(rng.randint(0, i))]); continue
            nrm = np.array([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)]); nrm /= np.linalg.norm(nrm)
            contacts.append((tuple(np.array([rng.uniform(-0.7, 0.7), 0.3 + rng.uniform(-0.02, 0.02), rng.uniform(-2, 2)], dtype=np.float32)),
                             tuple(nrm.astype(np.float32)), float(np.float32(rng.uniform(0, 0.03))), int(rng.randint(0, 2))))
        lv = tuple(rng.uniform(-3, 3, 3).astype(np.float32)); av = tuple(rng.uniform(-1, 1, 3).astype(np.float32))
        r = _unit(oracle, 1000.0, (1.4, 1.35, 4.7), (0, 1, 0), lv, av, contacts)
        worst_iter = max(worst_iter, r['it'])
        lam, lo, hi = r['lam'], r['lo'], r['hi']
        assert np.all(lam[0::3] >= 0.0)
        assert np.all(lam >= lo - 1e-3) and np.all(lam[1::3] <= hi[1::3] + 1e-3) and np.all(lam[2::3] <= hi[2::3] + 1e-3)
        # float64 rebuild: S x = b + w on a lone body (K = M^-1)
        m = 1000.0
        I = np.array([m / 12 * (1.35 ** 2 + 4.7 ** 2), m / 12 * (1.4 ** 2 + 4.7 ** 2), m / 12 * (1.4 ** 2 + 1.35 ** 2)])
        Minv = np.diag([1 / m] * 3 + list(1 / I))
        rows, cvec, cfm = [], [], []
        for (p, nrm, depth, kind) in contacts:
            nv = np.array(nrm, dtype=np.float64); c1 = np.array(p, dtype=np.float64) - np.array([0, 1, 0.0])
            if abs(nv[2]) > np.sqrt(0.5):
                a = nv[1] ** 2 + nv[2] ** 2; k = 1 / np.sqrt(a); t1 = np.array([0, -nv[2] * k, nv[1] * k]); t2 = np.array([a * k, -nv[0] * t1[2], nv[0] * t1[1]])
            else:
                a = nv[0] ** 2 + nv[1] ** 2; k = 1 / np.sqrt(a); t1 = np.array([-nv[1] * k, nv[0] * k, 0]); t2 = np.array([-nv[2] * t1[1], nv[2] * t1[0], a * k])
            erp, scfm, bounce = ((0.3, 1e-4, 0.01), (0.714285731, 0.000952380942, 0.0))[kind]
            for d in (nv, t1, t2):
                rows.append(np.concatenate([d, np.cross(c1, d)]))
            out = np.dot(np.cross(c1, nv), av) + np.dot(nv, lv)
            c = min(float(FPS) * erp * depth, 3.0)
            if -out > 0: c = max(c, bounce * -out)
            cvec += [c, 0, 0]; cfm += [scfm, 1e-7, 1e-7]
        J = np.array(rows); fps = float(FPS)
        S = J @ Minv @ J.T + np.diag(np.array(cfm) * fps)
        tmp1 = np.concatenate([np.array(lv) * fps + np.array([0, -9.80665, 0]), np.array(av) * fps])
        b = np.array(cvec) * fps - J @ tmp1
        w = S @ lam - b
        scale = np.abs(b).max() + 1.0
        for i in range(3 * n):
            free = lo[i] + 1e-6 * scale < lam[i] < hi[i] - 1e-6 * scale
            if free:
                assert abs(w[i]) < 2e-3 * scale, (trial, i, w[i], scale)
            elif lam[i] <= lo[i] + 1e-6 * scale and lo[i] < hi[i]:
                assert w[i] > -2e-3 * scale, (trial, i)
            elif hi[i] > lo[i]:
                assert w[i] < 2e-3 * scale, (trial, i)
    assert worst_iter <= 40   # both stages together; the cap is 64 per stage


@pytest.fixture(scope='module')
def walled(hostlib):
    import synthetic_tracks, tempfile
    d = tempfile.mkdtemp(prefix='pdb_walled_')
    synthetic_tracks.make_base(d, tracks=('walled',))
    return pc.build_track(hostlib, d, 'walled')


def _kinetic(S, P):
    e = 0.0
    for b in range(P.numBodies):
        m = P.bodies[b].mass
        v = np.array(S.body[b].lvel[:]); w = np.array(S.body[b].avel[:]); R = np.array(S.body[b].R[:]).reshape(3, 3)
        Iw = R @ np.diag(P.bodies[b].inertia[:]) @ R.T
        e += 0.5 * m * v @ v + 0.5 * w @ Iw @ w
    return e


@pytest.mark.parametrize('steer,gas,shift,speed', [(0.0, -1.0, 65.0, 15.0), (0.0, 1.0, 65.0, 15.0), (0.06, 0.6, 0.0, 0.0), (-0.05, 1.0, 30.0, 10.0)])
def test_drive_invariants_on_the_walled_strip(oracle, hostlib, walled, steer, gas, shift, speed):
    """every tick with live contact joints: lambda_n >= 0, friction inside its box; the chassis stays between the side walls
    (+-7.5 m) and short of the wall across the road (z = -120); the hit does not create kinetic energy"""
    P = car_params(AE86)
    s0 = pc.DynState()
    assert hostlib.pdb_initial_state(C.byref(P), walled, C.byref(s0)) == 0
    for b in range(P.numBodies):
        s0.body[b].pos[2] += shift; s0.body[b].lvel[2] = speed
    h = oracle.cpuref_create(C.byref(P), walled, len(walled), C.byref(s0))
    S = pc.DynState()
    lam = (C.c_float * 96)(); lo = (C.c_float * 96)(); hi = (C.c_float * 96)(); it = C.c_int()
    contact_ticks = 0; e_prev = _kinetic(s0, P); e_before_hit = None; e_after_max = 0.0; first = None
    for t in range(2600):
        oracle.cpuref_step_env(h, steer, gas)
        oracle.cpuref_get_state(h, C.byref(S))
        e = _kinetic(S, P)
        if S.numContacts > 0:
            contact_ticks += 1
            n = oracle.cpuref_last_contact_rows(h, lam, lo, hi, 96, C.byref(it))
            assert n == 3 * S.numContacts and it.value <= 512
            L = np.array(lam[:n]); LO = np.array(lo[:n]); HI = np.array(hi[:n])
            assert np.all(L[0::3] >= 0.0)
            assert np.all(L >= LO) and np.all(L[1::3] <= HI[1::3]) and np.all(L[2::3] <= HI[2::3])
            if first is None:
                first = t; e_before_hit = e_prev
        if first is not None and t - first < 120:
            e_after_max = max(e_after_max, e)
        e_prev = e
        x, z = S.body[0].pos[0], S.body[0].pos[2]
        assert abs(x) < 7.5 and z < -120.0, (t, x, z)
        assert np.isfinite(e)
    oracle.cpuref_destroy(h)
    assert contact_ticks > 5
    # the hit takes kinetic energy out; what the joints give back is the push-out of the penetration, capped at
    # contactMaxCorrectingVel = 3 m/s (1/2 m 3^2 = 5 kJ for this car), plus what the engine adds in a third of a second
    assert e_after_max < e_before_hit + 5.0e3 + 150e3 * 120 * float(H), (e_before_hit, e_after_max)
