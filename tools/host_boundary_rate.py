#!/usr/bin/env python3
"""PCIe-inclusive rate of the synchronous host boundary (pdb_step_host: actions H2D, one tick, outputs D2H, sync) --
the figure DESIGN.md quotes next to the HBM-resident bench value; never the bench `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import numpy as np, pdbatch, sharding
for n in (1, 64, 4096, 16384):
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
    a = sharding.global_actions(n, 1234)
    for _ in range(50): b.step_host(a)
    t0 = time.perf_counter(); k = 300
    for _ in range(k): b.step_host(a)
    dt = time.perf_counter() - t0
    print('pdb_step_host n=%6d: %.1f us per tick, %.3f M env-steps/s (PCIe-inclusive, synchronous)' % (n, dt / k * 1e6, n * k / dt / 1e6))
    b.close()
