#!/usr/bin/env python3
"""Diagnostic (GPU box): the headline workload (16384 cars on ek_akina, scripted law, env loop) stepped one launch per tick for thousands of ticks; every 100 ticks:
wall time per tick, cars in the contact pass, episode ends, off-track / stuck / collision flags, non-finite poses, the spread of the cars' heights."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import torch
import pdbatch, pdb_ctypes as pc, projectd_env as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
P = pdbatch.packed_params(); trk = pdbatch.reference_track('ek_akina')
b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
b.set_seed(np.arange(n, dtype=np.uint32) * 2654435761 % 4294967291 + 1)
b.reset(mode=2); b.set_env(E.EnvConfig(teleport_mode=2))
dev = 'cuda:0'
class _Arr:
    def __init__(self, ptr, shape): self.__cuda_array_interface__ = {'shape': shape, 'typestr': '<f4', 'data': (ptr, False), 'version': 2}
out_t = torch.as_tensor(_Arr(b.out_device_ptr(), (n, 26)), device=dev); act_t = torch.as_tensor(_Arr(b.actions_device_ptr(), (n, 2)), device=dev)
phi = torch.from_numpy(np.random.RandomState(2345).uniform(0.0, 2.0 * np.pi, n).astype(np.float32)).to(dev)
b.set_stream(torch.cuda.current_stream().cuda_stream)
acc = torch.zeros(6, device=dev, dtype=torch.int64)
t0 = time.perf_counter()
for t in range(ticks):
    b.step_async()
    o, a = out_t, act_t
    torch.add(o[:, 21], o[:, 20], alpha=-1.0, out=a[:, 0]).mul_(0.03).add_(o[:, 12], alpha=-1.0).add_(o[:, 4], alpha=0.15)
    a[:, 0].clamp_(-1.0, 1.0)
    torch.sin(phi + (2.0 * np.pi / 7.0) * (t / 333.0), out=a[:, 1]); a[:, 1].mul_(0.4 / 0.45).add_(0.5 / 0.45 - 1.0)
    fl = o[:, 25].view(torch.int32)
    for k, bit in enumerate((1, 2, 4, 8, 16, 32)):
        acc[k] += ((fl & bit) != 0).sum()
    if (t + 1) % 100 == 0:
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / 100 * 1e6
        st = b.get_state(0, 2048)
        raw = np.frombuffer(st, dtype=np.uint8).reshape(2048, C.sizeof(pc.DynState))
        off = pc.DynState.body.offset
        pos = raw[:, off:off + 12].copy().view(np.float32)
        a6 = acc.cpu().numpy() / 100.0; acc.zero_()
        print('tick %5d: %7.1f us/tick | in pass %4d | per tick: hit %.1f off %.1f stuck %.1f term %.1f reset %.1f fault %.1f | y of 2048 cars: min %.1f max %.1f nonfinite %d' % (
            t + 1, dtm, b.lib.pdb_contact_pass_load(b.h, 4), a6[0], a6[1], a6[2], a6[3], a6[4], a6[5], np.nanmin(pos[:, 1]), np.nanmax(pos[:, 1]), int((~np.isfinite(pos)).sum())), flush=True)
        t0 = time.perf_counter()
b.close()
