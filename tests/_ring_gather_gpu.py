"""Worker for tests/test_gpu_parity.py::test_ring_gather_holds_every_tick (run as a subprocess: it initialises a one-rank RCCL group).  Free-running
partitions write k-tick trajectory rings in place, started WITHOUT waiting for the batch's stream (pdb_step_ring fork=False); sharding.TrajectoryGather
all-gathers every full ring on the current stream and orders a ring's reuse on the partitions' own streams.  Every gathered ring must hold exactly
the [n, 26] blocks a plain batch produces for those ticks."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))


def main():
    import torch, torch.distributed as dist
    torch.cuda.init()
    import pdbatch, sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', init_method='env://')
    n, k, rings, parts = 900, 8, 9, 3
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('touge')
    acts = sharding.global_actions(n, 21)
    a = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    ref = np.zeros((rings * k, n, 26), np.float32)
    for t in range(rings * k):
        o = a.step_host(acts)
        ref[t, :, :24] = o['obs']; ref[t, :, 24] = o['reward']; ref[t, :, 25].view(np.int32)[:] = o['flags']
    b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    b.set_stream(torch.cuda.current_stream().cuda_stream)
    b.upload_actions(acts)
    b.set_partitions(parts)
    streams = [torch.cuda.ExternalStream(b.partition_stream(p), device='cuda:0') for p in range(parts)]
    g = sharding.TrajectoryGather(n, 1, 'cuda:0', dist, k=k, force=True, producer_wait=b.wait_partitions, producer_streams=lambda: streams)
    g.warm()
    got = []
    for r in range(rings):
        t = r * k
        # uneven chunks inside the ring, as bench.py's run() enqueues them
        done = 0
        while done < k:
            m = min(3, k - done)
            b.step_ring(m, g.ring(t + done).data_ptr(), k, (t + done) % k, join=False, fork=False)
            done += m
        torch.cuda._sleep(1500000)     # the gather's stream is late (as a gather over xGMI is): the partitions run ahead and have to be held at the ring they would overwrite
        full = g.after_tick(t + k - 1)
        assert full is not None
        g.work[r & 1].wait()           # the current stream waits for this ring's collective (it runs on the process group's own stream) ...
        got.append(full[0].clone())    # ... and takes what it delivered
    g.finish(); torch.cuda.synchronize()
    got = np.concatenate([x.cpu().numpy() for x in got], 0)
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    sa, sb = bytes(a.get_state()), bytes(b.get_state())
    a.close(); b.close()
    dist.destroy_process_group()
    print('RING_GATHER', 'OK' if ok and sa == sb else 'MISMATCH', flush=True)
    sys.exit(0 if ok and sa == sb else 1)


if __name__ == '__main__':
    main()
