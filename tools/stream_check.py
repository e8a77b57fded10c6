#!/usr/bin/env python3
"""Diagnostic (GPU box): do torch's kernels follow torch.cuda.set_stream(ExternalStream(partition stream))?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import torch, numpy as np
import pdbatch
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
b = pdbatch.Batch(4096, P, trk, device=0, action_mode=1)
b.set_partitions(3)
stream = torch.cuda.current_stream()
b.set_stream(stream.cuda_stream)
ps = [torch.cuda.ExternalStream(b.partition_stream(p), device='cuda:0') for p in range(3)]
print('current at start', hex(stream.cuda_stream), 'partition streams', [hex(s.cuda_stream) for s in ps])
for p in range(3):
    torch.cuda.set_stream(ps[p])
    print('after set_stream(%d): current == partition stream: %s' % (p, torch.cuda.current_stream().cuda_stream == ps[p].cuda_stream))
    torch.cuda._sleep(400000000)
    y = torch.zeros(1024, device='cuda:0'); y.fill_(1.0)
    print('   busy right after a long sleep kernel: partition stream %s, the default stream %s' % (not ps[p].query(), not torch.cuda.default_stream().query()))
    torch.cuda.synchronize()
torch.cuda.set_stream(stream)
b.close()
