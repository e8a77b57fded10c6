"""Worker for tests/test_sharding.py: one rank of a world_size-2 gloo job.  The stepper on each rank is the CPU oracle
(test infrastructure) standing in for the GPU batch -- what is under test is the host-side sharding/gather logic of
projectd-core_amd/sharding.py, which is the code bench.py runs over RCCL."""
import ctypes as C, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))


def step_block(orc, handles, actions):
    import pdb_ctypes as pc
    out = np.zeros((len(handles), 26), dtype=np.float32)
    so = pc.StepOut()
    for i, h in enumerate(handles):
        orc.cpuref_step_env(h, float(actions[i, 0]), float(actions[i, 1]))
        orc.cpuref_get_out(h, C.byref(so))
        out[i, :24] = so.obs[:]
        out[i, 24] = so.reward
        out[i, 25:26].view(np.int32)[0] = so.flags
    return out


def main(rank, world, port, n_global, ticks, out_path):
    import torch, torch.distributed as dist
    import pdb_ctypes as pc, oracle_ctypes, pdbatch, sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='env://')
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(host_only=True); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    first, last = sharding.shard_bounds(n_global, world, rank)
    n_local = last - first
    # learner (rank 0) owns the global actions and scatters each rank's slice
    all_a = torch.from_numpy(sharding.global_actions(n_global, 1234)) if rank == 0 else None
    mine = sharding.scatter_actions(all_a, n_local, world, rank, 'cpu', dist).numpy()
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n_local)]
    k = 8
    gather = sharding.TrajectoryGather(n_local, world, 'cpu', dist, k=k)
    g = None
    for t in range(ticks):
        gather.write(t, torch.from_numpy(step_block(orc, hs, mine)))      # stands in for the kernel writing the ring slot
        full = gather.after_tick(t)
        if full is not None:
            g = full.clone()                                                # [world, k, n_local, 26]
    gather.finish()
    slow = sharding.max_over_ranks(1.0 + rank, 'cpu', dist, world)
    if rank == 0:
        np.save(out_path, np.concatenate([g.numpy().reshape(-1), [slow]]))
    for h in hs:
        orc.cpuref_destroy(h)
    dist.barrier()
    dist.destroy_process_group()


def exchange_actions(t, n_global):
    """the learner's actions of tick t, by global car id: they change every tick, so a scatter that delivered stale or misplaced rows shows"""
    i = np.arange(n_global, dtype=np.float64)
    a = np.empty((n_global, 2), np.float32)
    a[:, 0] = 0.3 * np.sin(0.37 * i + 0.011 * t)
    a[:, 1] = np.cos(0.23 * i + 0.007 * t)
    return a


def part_ranges(n_local, parts):
    b = [n_local * p // parts for p in range(parts + 1)]
    return [(b[p], b[p + 1] - b[p]) for p in range(parts)]


def main_exchange(rank, world, port, n_global, ticks, out_path, parts=2):
    """sharding.PartitionExchange as bench.py drives it: per tick and partition, scatter of the partition's action rows from the learner ->
    the partition's tick (the oracle stands in for the kernel) -> all-gather of its output rows.  Rank 0 saves every tick's gathered blocks."""
    import torch, torch.distributed as dist
    import pdb_ctypes as pc, oracle_ctypes, pdbatch, sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', init_method='env://')
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(host_only=True); orc = oracle_ctypes.load_oracle(True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    first, last = sharding.shard_bounds(n_global, world, rank)
    n_local = last - first
    rng = part_ranges(n_local, parts)
    exch = sharding.PartitionExchange(rng, world, rank, 'cpu', dist)
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n_local)]
    act = torch.zeros((n_local, 2), dtype=torch.float32)
    out = torch.zeros((n_local, 26), dtype=torch.float32)
    hist = np.zeros((ticks, world, n_local, 26), np.float32)
    for t in range(ticks):
        if rank == 0:
            exch.load_actions(torch.from_numpy(exchange_actions(t, n_global)))
        for p, (f, c) in enumerate(rng):
            exch.scatter(p, act[f:f + c])
            out[f:f + c] = torch.from_numpy(step_block(orc, hs[f:f + c], act[f:f + c].numpy()))
            exch.gather(p, out[f:f + c])
            hist[t][:, f:f + c] = exch.gathered[p].numpy()
    if rank == 0:
        np.save(out_path, hist)
    for h in hs:
        orc.cpuref_destroy(h)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    if sys.argv[1] == 'exchange':
        main_exchange(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7])
    else:
        main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
