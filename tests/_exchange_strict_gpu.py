"""Worker for tests/test_gpu_parity.py::test_multi_rank_code_paths_with_one_rccl_rank (a subprocess: it initialises a one-rank RCCL group).  Everything
bench.py and sharding.py do with N > 1 ranks, executed by ONE rank through the exact calls N ranks make (strict=True: no world == 1 shortcut):
  - sharding.max_over_ranks (all-reduce MAX), sharding.scatter_actions (dist.scatter through the backend);
  - sharding.LibraryExchange: id broadcast, agreement all-reduce, pdb_comm_init, then per partition and tick the library's scatter -> tick -> all-gather;
  - sharding.PartitionExchange: a process group per partition, dist.scatter + all_gather_into_tensor on the partition's stream;
  - sharding.TrajectoryGather behind per-partition closed loops (the headline's N > 1 form): the kernels write the ring slot in place, a ring's gather waits
    for every partition's last kernel of it, a ring's reuse is ordered on the partitions' streams.
Every gathered block equals what a plain batch stepped with the same actions produces; the final records agree."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))


def rows(o):
    r = np.zeros((len(o), 26), np.float32)
    r[:, :24] = o['obs']; r[:, 24] = o['reward']; r[:, 25].view(np.int32)[:] = o['flags']
    return r


def main():
    import torch, torch.distributed as dist
    torch.cuda.init()
    import pdbatch, sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', init_method='env://')
    dev = 'cuda:0'
    n, ticks, parts, k = 600, 48, 3, 8
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('touge')
    assert sharding.max_over_ranks(1.25, dev, dist, 1, strict=True) == 1.25
    rng = np.random.RandomState(4)
    acts = [np.stack([rng.uniform(-0.4, 0.4, n), rng.uniform(-1, 1, n)], 1).astype(np.float32) for _ in range(ticks)]
    ok = True

    def fresh():
        b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
        b.set_stream(torch.cuda.current_stream().cuda_stream)
        b.set_partitions(parts)
        return b, [b.partition_range(p) for p in range(parts)], [torch.cuda.ExternalStream(b.partition_stream(p), device=dev) for p in range(parts)]

    class _Arr:
        def __init__(self, ptr, shape): self.__cuda_array_interface__ = {'shape': shape, 'typestr': '<f4', 'data': (ptr, False), 'version': 2}

    # reference: a plain batch
    a = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    ref = [rows(a.step_host(acts[t])) for t in range(ticks)]
    sa = bytes(a.get_state()); a.close()

    # 1. the library's own exchange
    b, rg, st = fresh()
    ex = sharding.LibraryExchange(b, rg, 1, 0, dev, dist, strict=True)
    for t in range(ticks):
        mine = sharding.scatter_actions(torch.from_numpy(acts[t]).to(dev), n, 1, 0, dev, dist, strict=True)   # (dist.scatter to itself)
        ex.load_actions(mine); torch.cuda.synchronize()
        for p in range(parts):
            ex.step(p)
        b.wait_partitions(); torch.cuda.synchronize()
        for p, (f, c) in enumerate(rg):
            ok = ok and np.array_equal(ex.gathered[p][0].cpu().numpy().view(np.uint32), ref[t][f:f + c].view(np.uint32))
    ok = ok and bytes(b.get_state()) == sa
    b.close()
    print('library exchange', ok, flush=True)

    # 2. the same through torch.distributed, a process group per partition
    b, rg, st = fresh()
    out_t = torch.as_tensor(_Arr(b.out_device_ptr(), (n, 26)), device=dev); act_t = torch.as_tensor(_Arr(b.actions_device_ptr(), (n, 2)), device=dev)
    px = sharding.PartitionExchange(rg, 1, 0, dev, dist, strict=True)
    for t in range(ticks):
        px.load_actions(torch.from_numpy(acts[t]).to(dev)); torch.cuda.synchronize()
        for p, (f, c) in enumerate(rg):
            with torch.cuda.stream(st[p]):
                px.scatter(p, act_t[f:f + c])
                b.step_partition(p, out_t.data_ptr())
                px.gather(p, out_t[f:f + c], wait=True)
        b.wait_partitions(); torch.cuda.synchronize()
        for p, (f, c) in enumerate(rg):
            ok = ok and np.array_equal(px.gathered[p][0].cpu().numpy().view(np.uint32), ref[t][f:f + c].view(np.uint32))
    ok = ok and bytes(b.get_state()) == sa
    b.close()
    print('partition exchange', ok, flush=True)

    # 3. trajectory rings behind per-partition loops (a per-tick "policy" on each partition's stream: the next tick's actions into its rows)
    b, rg, st = fresh()
    act_t = torch.as_tensor(_Arr(b.actions_device_ptr(), (n, 2)), device=dev)
    dacts = [torch.from_numpy(x).to(dev) for x in acts]
    g = sharding.TrajectoryGather(n, 1, dev, dist, k=k, force=True, producer_wait=b.wait_partitions, producer_streams=lambda: st)
    g.warm()
    act_t.copy_(dacts[0]); torch.cuda.synchronize()
    got = []
    for t in range(ticks):
        o = g.slot(t)
        for p, (f, c) in enumerate(rg):
            with torch.cuda.stream(st[p]):
                b.step_partition(p, o.data_ptr())
                if t + 1 < ticks:
                    act_t[f:f + c].copy_(dacts[t + 1][f:f + c])
        full = g.after_tick(t)
        if full is not None:
            g.work[(t // k) & 1].wait()
            got.append(full[0].clone())
    g.finish(); torch.cuda.synchronize()
    got = np.concatenate([x.cpu().numpy() for x in got], 0)
    ok = ok and np.array_equal(got.view(np.uint32), np.stack(ref).view(np.uint32)) and bytes(b.get_state()) == sa
    b.close()
    print('ring gather behind partition loops', ok, flush=True)
    dist.destroy_process_group()
    print('STRICT_PATHS', 'OK' if ok else 'MISMATCH', flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
