"""The N > 1 path on CPU: 2, 4 and 8 gloo ranks, each stepping its contiguous block of cars, action
scatter from the learner rank, gather of 8-tick trajectory rings of [n,26] output blocks -- must give exactly what one process gives
for all cars (results keyed by global car id, invariant to the number of ranks)."""
import ctypes as C, os, socket, subprocess, sys, tempfile
import numpy as np
import pytest
import pdb_ctypes as pc
import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_and_partition():
    for n in (1, 7, 8, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [y - x for x, y in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(8, 2, 2)


def test_global_actions_are_invariant_to_the_sharding():
    a = sharding.global_actions(8192, 1234)
    assert a.dtype == np.float32 and a.shape == (8192, 2)
    assert np.all(np.abs(a[:, 0]) <= 0.3) and np.all(np.abs(a[:, 1]) <= 1.0)
    f, l = sharding.shard_bounds(8192, 2, 1)
    assert np.array_equal(sharding.global_actions(8192, 1234)[f:l], a[4096:])


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 4, 8])
def test_gather_over_ranks_equals_single_process(built, oracle, world):
    """world gloo ranks ({1, 2, 4, 8} shards), 16 cars: the gathered last ring equals the single-process one"""
    import _sharding_worker as w
    import pdbatch, oracle_ctypes
    n_global, ticks = 16, 40
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'gathered.npy')
        env = dict(os.environ, OMP_NUM_THREADS='1')
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_sharding_worker.py'), str(r), str(world), str(port), str(n_global), str(ticks), out], env=env)
                 for r in range(world)]
        rcs = [p.wait(timeout=540) for p in procs]
        assert rcs == [0] * world
        got = np.load(out)
    assert got[-1] == float(world)                          # max over ranks of (1 + rank)
    got = got[:-1].astype(np.float32).reshape(world, 8, n_global // world, 26)    # [world, k, n_local, 26]: the last full 8-tick ring
    # single process, all cars
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(host_only=True); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n_global)]
    a = sharding.global_actions(n_global, 1234)
    ring = np.zeros((8, n_global, 26), np.float32)
    for t in range(ticks):
        ring[t % 8] = w.step_block(orc, hs, a)
    for h in hs:
        orc.cpuref_destroy(h)
    ref = ring.reshape(8, world, n_global // world, 26).transpose(1, 0, 2, 3)      # rank-major like the gathered tensor
    assert np.array_equal(got.view(np.int32), np.ascontiguousarray(ref).view(np.int32))
    assert np.any(ref[..., :24] != 0)


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 4])
def test_partition_exchange_over_ranks_equals_single_process(built, oracle, world):
    """sharding.PartitionExchange with world > 1 (it had never run): every tick the learner's actions -- different every tick -- are
    scattered per partition, the partition steps, its output rows are all-gathered; what rank 0 holds after every tick equals a single process
    stepping all cars with those actions, bit for bit"""
    import _sharding_worker as w
    import pdbatch, oracle_ctypes
    n_global, ticks = 16, 30
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'hist.npy')
        env = dict(os.environ, OMP_NUM_THREADS='1')
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_sharding_worker.py'), 'exchange', str(r), str(world), str(port), str(n_global), str(ticks), out], env=env)
                 for r in range(world)]
        rcs = [p.wait(timeout=540) for p in procs]
        assert rcs == [0] * world
        got = np.load(out)                                   # [ticks, world, n_local, 26]
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(host_only=True); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n_global)]
    ref = np.zeros((ticks, n_global, 26), np.float32)
    for t in range(ticks):
        ref[t] = w.step_block(orc, hs, w.exchange_actions(t, n_global))
    for h in hs:
        orc.cpuref_destroy(h)
    assert np.array_equal(got.reshape(ticks, n_global, 26).view(np.int32), ref.view(np.int32))
    assert np.any(ref[..., :24] != 0)
