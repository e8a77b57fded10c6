"""Multi-car simulators on the device (pdb_set_world_size): worlds of G consecutive lanes coupled through the slipstream (reference Car::updateAirPressure, Car.cpp:557-585;
Sim/SlipStream.cpp; the ERP switch only a simulator's first car takes, Car.cpp:426).  The oracle side of every comparison is held to the reference's own translation units
with two cars in ONE Simulator by the twocar_* goldens (tests/test_oracle_golden.py)."""
import ctypes as C, os
import numpy as np
import pytest
import pdb_ctypes as pc
import oracle_ctypes
from conftest import car_params


def _world_run(G, worlds, ticks, partitions=None, model='pdb_slip_ae86', check_every=50, seed=5, track='flat', steer=0.01):
    import pdbatch, parity_util
    n = G * worlds
    P = car_params(model); trk = pdbatch.synthetic_track(track)
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(portable_math=True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    init = (pc.DynState * n)()
    for w in range(worlds):
        for c in range(G):   # the cars of a world a few car lengths apart along the line, the worlds spread along the straight
            s = pc.DynState.from_buffer_copy(bytes(S0))
            assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(0.02 * w + 0.0034 * c), C.byref(s)) == 0
            s.randState = 1 + 31 * (w * G + c)
            C.memmove(C.byref(init[w * G + c]), C.byref(s), C.sizeof(s))
    b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    b.set_world_size(G)
    if partitions:
        b.set_partitions(partitions)
    b.set_state(init)
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(init[i])) for i in range(n)]
    for i in range(n):
        orc.cpuref_set_guid(hs[i], i % G)
    rs = np.random.RandomState(seed)
    acts = np.zeros((n, 2), np.float32)
    acts[:, 0] = rs.uniform(-steer, steer, n)
    acts[:, 1] = np.where(np.arange(n) % G == 0, 1.0, rs.uniform(-0.6, 0.2, n))   # a world's first car (the one behind) flat out, the others slower: it runs through their wakes
    slips = (pc.SlipState * n)(); others = (pc.SlipState * max(1, G - 1))()
    thinned = 0
    try:
        for t in range(ticks):
            b.step_host(acts)
            for i in range(n):
                orc.cpuref_get_slip(hs[i], C.byref(slips[i]))
            for i in range(n):
                w0 = (i // G) * G
                k = 0
                for o in range(w0, w0 + G):   # the simulator's car order, the car itself left out
                    if o != i:
                        C.memmove(C.byref(others[k]), C.byref(slips[o]), C.sizeof(pc.SlipState)); k += 1
                orc.cpuref_set_other_slips(hs[i], others, G - 1)
            for i in range(n):
                orc.cpuref_step_env(hs[i], float(acts[i, 0]), float(acts[i, 1]))
            if (t + 1) % check_every == 0 or t == ticks - 1:
                st = b.get_state(); sl = b.get_slipstreams()
                for i in range(n):
                    so = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(so))
                    rel, name, vg, vc, bad_int = parity_util.compare_states(st[i], so)
                    assert not bad_int and rel == 0.0, (t, i, name, vg, vc, bad_int[:4])
                    ss = pc.SlipState(); orc.cpuref_get_slip(hs[i], C.byref(ss))
                    assert bytes(ss) == bytes(sl[st[i].simFrame & 1][i]), (t, i)
        return b, hs, orc
    except Exception:
        b.close()
        for h in hs:
            orc.cpuref_destroy(h)
        raise


@pytest.mark.gpu
@pytest.mark.parametrize('G,worlds,partitions', [(2, 24, None), (2, 36, 3), (3, 8, None), (4, 3, None)])
def test_worlds_equal_the_oracle_car_by_car(built, G, worlds, partitions):
    """batches of several worlds (two, three and four cars each; cut into three free-running partitions on whole worlds): every car's record and wake, every 50 ticks of
    1500, bit for bit the oracle's -- a world's first car closes in on the others through their wakes"""
    b, hs, orc = _world_run(G, worlds, 1500, partitions)
    b.close()
    for h in hs:
        orc.cpuref_destroy(h)


@pytest.mark.gpu
def test_worlds_through_the_contact_pass(built):
    """two-car worlds on the walled strip, steered into the side walls and the cross wall: the cars that touch something take their ticks in the contact pass (one
    kernel or the pair), the others in the first pass -- whichever pass stores a car's record leaves its wake too; records, live contact counts and wakes bit for bit"""
    b, hs, orc = _world_run(2, 12, 1800, model='ks_toyota_ae86_drift', track='walled', steer=0.12, check_every=30)
    try:
        st = b.get_state()
        assert sum(1 for i in range(24) if st[i].damageZoneLevel[4] > 0) >= 6   # cars really did hit the walls
    finally:
        b.close()
        for h in hs:
            orc.cpuref_destroy(h)


@pytest.mark.gpu
def test_a_snapshot_with_its_wakes_replays_bit_for_bit(built):
    """records + contact joints + the wakes' two buffers (pdb_get_slipstreams) are the whole state of a batch of multi-car simulators: restored, the next 300 ticks are the
    same bytes"""
    import pdbatch
    b, hs, orc = _world_run(2, 8, 400)
    for h in hs:
        orc.cpuref_destroy(h)
    try:
        snap_s = bytes(b.get_state()); sl = b.get_slipstreams(); snap_sl = [bytes(sl[0]), bytes(sl[1])]
        acts = np.zeros((16, 2), np.float32); acts[:, 1] = np.where(np.arange(16) % 2 == 0, 1.0, -0.3)
        for _ in range(300):
            b.step_host(acts)
        end_a = bytes(b.get_state()); sla = b.get_slipstreams(); end_sl_a = [bytes(sla[0]), bytes(sla[1])]
        b.set_state((pc.DynState * 16).from_buffer_copy(snap_s))
        b.set_slipstreams([(pc.SlipState * 16).from_buffer_copy(snap_sl[0]), (pc.SlipState * 16).from_buffer_copy(snap_sl[1])])
        for _ in range(300):
            b.step_host(acts)
        slb = b.get_slipstreams()
        assert bytes(b.get_state()) == end_a and [bytes(slb[0]), bytes(slb[1])] == end_sl_a
    finally:
        b.close()


def test_world_size_is_refused_where_it_cannot_hold(built):
    """no device here: the entry points exist and refuse a null batch; the ABI test lists them"""
    lib = pc.load_product()
    assert lib.pdb_set_world_size(None, 2) != 0 and lib.pdb_get_slipstreams(None, 0, 0, None) != 0 and lib.pdb_set_slipstreams(None, 0, 0, None) != 0
