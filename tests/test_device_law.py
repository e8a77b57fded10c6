"""-m gpu: the device law (pdb_set_law, include/pdbatch.h) -- the next tick's steer / throttle evaluated by the tick's own launches from the observation
row they write -- against the same law evaluated on the HOST in float32 and fed to the CPU oracle tick by tick.  The law has no counterpart in the reference
(there the caller sets the controls every tick, PyProjectD.cpp:297-305); what is checked is that a batch stepping under it holds, bit for bit, the records,
observation rows and ACTIONS of cars stepped by the oracle with the host-evaluated law: through the first pass, the contact pass in both of its forms, free-running
partitions and a recorded graph."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import pytest

import pdb_ctypes as pc
import parity_util

pytestmark = pytest.mark.gpu


def law_host(obs, W, bias):
    """a[c] = bias[c] + sum_k obs[k] * W[k][c]: float32 products as the leaves 0..23 of a balanced binary tree over 32 slots (the kernel's waveSumF over the wave's
    lanes: the slots past 23 hold +0)"""
    obs = np.asarray(obs, np.float32); m = obs.shape[0]
    out = np.empty((m, 2), np.float32)
    for c in range(2):
        leaf = np.zeros((m, 32), np.float32)
        leaf[:, :24] = obs[:, :24] * W[None, :, c]
        while leaf.shape[1] > 1:
            leaf = leaf[:, 0::2] + leaf[:, 1::2]
        out[:, c] = bias[:, c] + leaf[:, 0]
    return out


def _download(ptr, nbytes):
    hip = C.CDLL('libamdhip64.so')
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    buf = np.empty(nbytes, np.uint8)
    assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), nbytes, 2) == 0
    return buf


def _weights(seed):
    r = np.random.RandomState(seed)
    W = (r.uniform(-1.0, 1.0, (24, 2)) * 0.004).astype(np.float32)            # every slot in play, small
    W[21, 0] += 0.03; W[20, 0] -= 0.03; W[12, 0] -= 1.0; W[4, 0] += 0.15      # bench.py's P-steer: side probes, the road's bend ahead, yaw damping
    W[2, 1] -= 0.05                                                           # throttle falls with the forward speed
    return W


@pytest.mark.parametrize('track,partitions,period,split', [('touge', 0, 0, None), ('touge', 3, 7, None), ('playground', 0, 5, '0'), ('playground', 3, 4, '1')])
def test_device_law_equals_the_host_fed_law(built, monkeypatch, track, partitions, period, split):
    import pdbatch, oracle_ctypes
    if split is not None:
        monkeypatch.setenv('PDB_CONTACT_SPLIT', split)       # the contact pass as one kernel / as the collide + resume pair
    n, ticks = 24, 700 if track == 'playground' else 400
    P = pdbatch.packed_params('ks_toyota_ae86_drift.env'); trk = pdbatch.synthetic_track(track)
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(portable_math=True)
    S0 = pc.DynState()
    assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    starts = (pc.DynState * n)()
    for i in range(n):
        s = pc.DynState.from_buffer_copy(bytes(S0))
        assert lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(i / n), C.byref(s)) == 0
        C.memmove(C.byref(starts[i]), C.byref(s), C.sizeof(s))
    W = _weights(11)
    r = np.random.RandomState(5)
    bias0 = np.array([0.01, 0.9], np.float32)
    table = None
    if period:
        table = np.empty((period, n, 2), np.float32)
        table[:, :, 0] = r.uniform(-0.05, 0.05, (period, n)); table[:, :, 1] = r.uniform(0.5, 1.0, (period, n))
    b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    b.set_state(starts)
    if partitions:
        b.set_partitions(partitions)
    b.set_law(W, bias0=bias0, table=table)
    hs = []
    for i in range(n):
        h = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)); orc.cpuref_set_state(h, C.byref(starts[i])); hs.append(h)
    pool = ThreadPoolExecutor(8)
    nb_out = C.sizeof(pc.StepOut)
    try:
        a = np.zeros((n, 2), np.float32)                     # tick 0 takes what the caller put into the action buffer
        b.upload_actions(a)
        oo = pc.StepOut()
        in_contact = 0
        t = 0
        while t < ticks:
            # one tick at a time (plain launches / the partitions' own streams), now and then five from the recorded graph
            m = 5 if (t % 50 == 20 and not partitions) else 1
            if partitions:
                b.step_ring(m, join=True)
            else:
                b.step(m)
            b.sync()
            for _ in range(m):
                list(pool.map(lambda i: orc.cpuref_step_env(hs[i], float(a[i, 0]), float(a[i, 1])), range(n)))
                obs = np.zeros((n, 24), np.float32)
                for i in range(n):
                    orc.cpuref_get_out(hs[i], C.byref(oo)); obs[i] = np.frombuffer(oo, dtype=np.float32, count=24)
                row = t % period if period else 0
                a = law_host(obs, W, table[row] if period else np.tile(bias0, (n, 1)))
                t += 1
            ag = _download(b.actions_device_ptr(), n * 8).view(np.float32).reshape(n, 2)
            assert np.array_equal(ag.view(np.uint32), a.view(np.uint32)), (t, np.argwhere(ag != a)[:4], ag[ag != a][:4], a[ag != a][:4])
            og = _download(b.out_device_ptr(), n * nb_out).view(np.dtype(pc.StepOut))
            assert np.array_equal(np.ascontiguousarray(og['obs']).view(np.uint32), obs.view(np.uint32)), t
            if t % 20 == 0 or t >= ticks:
                sg = b.get_state()
                for i in range(n):
                    sc = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(sc))
                    rel, name, vg, vc, bad_int = parity_util.compare_states(sg[i], sc)
                    assert not bad_int, (t, i, bad_int[:4])
                    assert rel == 0.0, (t, i, name, vg, vc)
                    assert sg[i].lawTick == (t % period if period else 0)
                    in_contact += 1 if sg[i].numContacts > 0 else 0
        sp = np.array([s.speed for s in b.get_state()])
        assert sp.max() > 1.0                                # the law's throttle got the cars going
        if track == 'playground':
            assert in_contact > 0                            # ... into the obstacles: the law's write-out ran in the contact pass too
    finally:
        b.close(); pool.shutdown()
        for h in hs:
            orc.cpuref_destroy(h)


def test_device_law_refusals_and_removal(built):
    import pdbatch
    P = pdbatch.packed_params('ks_toyota_ae86_drift.env'); trk = pdbatch.synthetic_track('flat')
    W = _weights(3)
    b = pdbatch.Batch(8, P, trk, device=0, action_mode=2)
    try:
        with pytest.raises(RuntimeError):
            b.set_law(W)                                     # eight controls a car: the law writes two
    finally:
        b.close()
    b = pdbatch.Batch(8, P, trk, device=0, action_mode=1)
    try:
        lib = b.lib
        assert lib.pdb_set_law(b.h, W.ctypes.data_as(C.c_void_p), None, W.ctypes.data_as(C.c_void_p), 0, 0) != 0   # a table without a period
        a = np.full((8, 2), 0.25, np.float32)
        b.upload_actions(a)
        b.set_law(W, bias0=np.array([0.0, 1.0], np.float32))
        b.step(3); b.sync()
        a1 = _download(b.actions_device_ptr(), 64).view(np.float32).reshape(8, 2)
        assert not np.array_equal(a1, a) and np.isfinite(a1).all()
        # a record whose row counter lies past the table's period (it came in through pdb_set_state): row 0, then on from there
        tab = np.random.RandomState(9).uniform(-0.2, 0.2, (3, 8, 2)).astype(np.float32)
        b.set_law(W, table=tab)
        st = b.get_state()
        for s_ in st:
            s_.lawTick = 99
        b.set_state(st)
        b.step(1); b.sync()
        og = _download(b.out_device_ptr(), 8 * C.sizeof(pc.StepOut)).view(np.dtype(pc.StepOut))
        want = law_host(np.ascontiguousarray(og['obs']), W, tab[0])
        got = _download(b.actions_device_ptr(), 64).view(np.float32).reshape(8, 2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert all(s_.lawTick == 1 for s_ in b.get_state())
        b.set_law(None)                                      # the caller's actions stand again
        b.upload_actions(a)
        b.step(3); b.sync()
        assert np.array_equal(_download(b.actions_device_ptr(), 64).view(np.float32).reshape(8, 2), a)
    finally:
        b.close()


def test_device_law_through_the_env_loop_in_the_kernel(built):
    """bench.py's headline regime: the law AND the env's episode rules (pdb_set_env: terminations, the reset tick's teleport to a random point and its zero action)
    inside the tick's launches.  Against a second batch with the same env mode that is FED the same law from the host (law_host on the rows pdb_step_host returns):
    every output row, flag and reward every tick, every record at the checks -- over many episode ends (half the cars creep and get `stuck` after 0.4 s, the others
    are steered into the walls of the strip), three free-running partitions on the law's side."""
    import pdbatch, projectd_env
    n, ticks, period = 48, 900, 5
    P = pdbatch.packed_params('ks_toyota_ae86_drift.env'); trk = pdbatch.synthetic_track('walled')
    W = np.zeros((24, 2), np.float32)
    W[20, 0] = -0.01; W[21, 0] = 0.01; W[2, 1] = -0.002                       # a weak feedback on top of the table
    r = np.random.RandomState(8)
    table = np.zeros((period, n, 2), np.float32)
    table[:, 0::2, 1] = -1.0                                                  # gas 0.1: creeps, no new track point within 0.4 s
    table[:, 1::2, 1] = 1.0; table[:, 1::2, 0] = np.where(np.arange(n // 2) % 2 == 0, 0.06, -0.06)[None, :]   # full throttle into a side wall
    table += r.uniform(-0.01, 0.01, table.shape).astype(np.float32)
    cfg = projectd_env.EnvConfig(teleport_mode=2, stuck_timeout=0.4, terminate_low_reward=-1.0e9)
    bl = pdbatch.Batch(n, P, trk, device=0, action_mode=1); bh = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    nb_out = C.sizeof(pc.StepOut)
    try:
        for b in (bl, bh):
            b.set_env(cfg)
        bl.set_partitions(3)
        bl.set_law(W, table=table)
        a = np.zeros((n, 2), np.float32)
        bl.upload_actions(a)
        ends = 0
        for t in range(ticks):
            bl.step_ring(1, join=True); bl.sync()
            oh = bh.step_host(a)
            ol = _download(bl.out_device_ptr(), n * nb_out).view(np.dtype(pc.StepOut))
            assert ol.tobytes() == np.asarray(oh).tobytes(), (t, np.argwhere(np.asarray(ol['flags']) != np.asarray(oh['flags']))[:4])
            ends += int(((np.asarray(oh['flags']) & 8) != 0).sum())
            a = law_host(np.ascontiguousarray(oh['obs']), W, table[t % period])
            assert np.array_equal(_download(bl.actions_device_ptr(), n * 8).view(np.uint32), a.view(np.uint32).ravel()), t
            if t % 50 == 49 or t == ticks - 1:
                sl, sh = bl.get_state(), bh.get_state()
                for i in range(n):
                    rel, name, vg, vc, bad_int = parity_util.compare_states(sl[i], sh[i])
                    assert not bad_int and rel == 0.0, (t, i, name, vg, vc, bad_int[:3])
                    assert sl[i].lawTick == (t + 1) % period and sh[i].lawTick == 0
        assert ends >= 40, ends
    finally:
        bl.close(); bh.close()
