#!/usr/bin/env python3
"""Diagnostic (GPU box): which streams share a hardware queue?  A batch steps rings on `parts` free-running partitions; an extra stream, created AFTER the
partitions' (as a process group's collective stream is), runs a CU-free 1 ms sleep kernel every 32 ticks.  If the rate drops, the extra stream shares a
hardware queue with a partition's stream (kernels of different streams in one queue run one after the other).  usage: hwqueue_probe.py <parts> [extra streams created before]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
torch.cuda.init()
import numpy as np, pdbatch, sharding
parts = int(sys.argv[1]); pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 4096
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
b.set_stream(torch.cuda.current_stream().cuda_stream)
b.upload_actions(sharding.global_actions(n, 1234))
early = [torch.cuda.Stream() for _ in range(pre)]      # streams that exist before the partitions' (never used)
b.set_partitions(parts)
def rate(extra, label):
    for _ in range(10):
        b.step_ring(32, join=False)
    b.wait_partitions(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(60):
        b.step_ring(32, join=False, fork=False)
        if extra is not None:
            with torch.cuda.stream(extra):
                torch.cuda._sleep(2400000)
    b.wait_partitions(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-70s %.2f M env-steps/s' % (label, n * 32 * 60 / dt / 1e6), flush=True)
rate(None, '%d partitions (+%d idle streams made before them), nothing else' % (parts, pre))
late = torch.cuda.Stream()
rate(late, '... + a 1 ms sleep per 32 ticks on a stream created after them')
late2 = torch.cuda.Stream()
rate(late2, '... on a second such stream')
rate(torch.cuda.current_stream(), '... on the current (null) stream')
b.close()
