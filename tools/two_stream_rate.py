"""Diagnostic: configs[1] as S independent batches on S streams (cars are independent, so nothing orders one batch's tick
against another's): does overlapping the kernels' ramp-up / drain phases and the inter-kernel gap pay?
Usage: python tools/two_stream_rate.py [cars_total] [streams]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import torch
torch.cuda.init()
import pdbatch, sharding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
acts = sharding.global_actions(n, 1234)
per = n // S
bs, streams = [], []
for i in range(S):
    st = torch.cuda.Stream()
    b = pdbatch.Batch(per, P, trk, device=0, action_mode=1)
    b.set_stream(st.cuda_stream)
    b.upload_actions(acts[i * per:(i + 1) * per])
    bs.append(b); streams.append(st)
for _ in range(333):
    for b in bs: b.step_async()
torch.cuda.synchronize()
K = 3000
t0 = time.perf_counter()
for _ in range(K):
    for b in bs: b.step_async()
torch.cuda.synchronize()
t1 = time.perf_counter()
print('%d cars as %d batches on %d streams: %.2f us per tick of all cars, %.2f M env-steps/s' % (n, S, S, (t1 - t0) / K * 1e6, per * S * K / (t1 - t0) / 1e6))
for b in bs: b.close()
