"""Diagnostic: env-steps/s of configs[1] when the ticks are replayed from a captured hipGraph (pdb_step_n) instead of being
launched one by one -- how much of a tick is launch gap.  Usage: python tools/graph_rate.py [cars] [ticks_per_graph]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import torch
torch.cuda.init()
import pdbatch, pdb_ctypes as pc, sharding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = int(sys.argv[2]) if len(sys.argv) > 2 else 100
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
b.upload_actions(sharding.global_actions(n, 1234))
b.step(g); b.step(g); b.step(g)
t0 = time.perf_counter()
reps = 30
for _ in range(reps): b.step(g)
t1 = time.perf_counter()
print('graph replay: %d cars, %d ticks per graph: %.2f us per tick, %.2f M env-steps/s' % (n, g, (t1 - t0) / (reps * g) * 1e6, n * reps * g / (t1 - t0) / 1e6))
for _ in range(300): b.step_async()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps * g): b.step_async()
torch.cuda.synchronize()
t1 = time.perf_counter()
print('single launches: %.2f us per tick, %.2f M env-steps/s' % ((t1 - t0) / (reps * g) * 1e6, n * reps * g / (t1 - t0) / 1e6))
b.close()
