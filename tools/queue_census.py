#!/usr/bin/env python3
"""Diagnostic (GPU box): who is in the contact pass?  Steps a workload one launch per tick, synchronously, and after every tick compares
the number of cars the contact pass held (pdb_contact_pass_load) with what the records say: cars with live contact joints before / after the
tick, cars on their reset tick, cars whose collision flag rose.  A car in the pass that owns no joint before or after is a false positive of the
first pass's superset test.  usage: queue_census.py [workload] [cars] [ticks] [policy]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import pdbatch, pdb_ctypes as pc, projectd_env as E

wl = sys.argv[1] if len(sys.argv) > 1 else 'ek_akina'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 600
policy = sys.argv[4] if len(sys.argv) > 4 else 'scripted'
settle = int(os.environ.get('SETTLE', 400))

P = pdbatch.packed_params()
trk = pdbatch.reference_track(wl) if wl in pdbatch.REFERENCE_TRACKS else pdbatch.synthetic_track(wl)
b = pdbatch.Batch(n, P, trk, device=0, action_mode=1)
b.set_seed(np.arange(n, dtype=np.uint32) * 2654435761 % 4294967291 + 1)
b.reset(mode=2)
b.set_env(E.EnvConfig(teleport_mode=2))
dev = 'cuda:0'


class _Arr:
    def __init__(self, ptr, shape, typestr='<f4'):
        self.__cuda_array_interface__ = {'shape': shape, 'typestr': typestr, 'data': (ptr, False), 'version': 2}


out_t = torch.as_tensor(_Arr(b.out_device_ptr(), (n, 26)), device=dev)
act_t = torch.as_tensor(_Arr(b.actions_device_ptr(), (n, 2)), device=dev)
phi = torch.from_numpy(np.random.RandomState(2345).uniform(0.0, 2.0 * np.pi, n).astype(np.float32)).to(dev)
stream = torch.cuda.current_stream()
b.set_stream(stream.cuda_stream)
dt = np.dtype(pc.DynState)
off_nc = pc.DynState.numContacts.offset


def num_contacts():
    st = b.get_state()
    raw = np.frombuffer(st, dtype=np.uint8).reshape(n, C.sizeof(pc.DynState))
    return raw[:, off_nc:off_nc + 4].copy().view(np.int32)[:, 0]


def policy_step(t):
    o, a = out_t, act_t
    if policy == 'scripted':
        torch.add(o[:, 21], o[:, 20], alpha=-1.0, out=a[:, 0]).mul_(0.03).add_(o[:, 12], alpha=-1.0).add_(o[:, 4], alpha=0.15)
        a[:, 0].clamp_(-1.0, 1.0)
        torch.sin(phi + (2.0 * np.pi / 7.0) * (t / 333.0), out=a[:, 1])
        a[:, 1].mul_(0.4 / 0.45).add_(0.5 / 0.45 - 1.0)
    else:
        a[:, 0] = (0.03 * (o[:, 21] - o[:, 20]) + 0.015 * (o[:, 19] - o[:, 18]) + 0.15 * o[:, 4]).clamp_(-1, 1)
        a[:, 1] = (0.3 * (12.0 - o[:, 2])).clamp_(-1, 1)


for t in range(settle):
    b.step_async(); policy_step(t)
torch.cuda.synchronize()
nc0 = num_contacts()
tot = dict(q=0, live=0, fresh=0, fp=0, reset=0, hit=0, term=0, off=0)
hist = []
for t in range(settle, settle + ticks):
    b.step_async()
    torch.cuda.synchronize()
    q = b.lib.pdb_contact_pass_load(b.h, 4)
    fl = out_t[:, 25].view(torch.int32).cpu().numpy()
    nc1 = num_contacts()
    live = int((nc0 > 0).sum()); fresh = int(((nc0 == 0) & (nc1 > 0)).sum()); owners = int(((nc0 > 0) | (nc1 > 0)).sum())
    tot['q'] += q; tot['live'] += live; tot['fresh'] += fresh; tot['fp'] += max(0, q - owners)
    tot['reset'] += int(((fl & 16) != 0).sum()); tot['hit'] += int(((fl & 1) != 0).sum()); tot['term'] += int(((fl & 8) != 0).sum()); tot['off'] += int(((fl & 2) != 0).sum())
    hist.append((q, owners, live, fresh))
    nc0 = nc1
    policy_step(t)
print('%s, %d cars, %d ticks, policy=%s' % (wl, n, ticks, policy))
print('per tick: in pass %.2f | joint owners before-or-after %.2f (live before %.2f, gained %.2f) | in pass without a joint (superset false positives) %.2f' %
      (tot['q'] / ticks, (tot['q'] - tot['fp']) / ticks, tot['live'] / ticks, tot['fresh'] / ticks, tot['fp'] / ticks))
print('per tick: reset ticks %.2f | terminated %.2f | collision flag %.2f | off-track flag %.2f' % (tot['reset'] / ticks, tot['term'] / ticks, tot['hit'] / ticks, tot['off'] / ticks))
odd = [h for i, h in enumerate(hist) if i % 2 == 0]; even = [h for i, h in enumerate(hist) if i % 2 == 1]
print('alternate ticks: in pass %.2f / %.2f' % (np.mean([h[0] for h in odd]), np.mean([h[0] for h in even])))
print('ticks with an empty pass: %d of %d' % (sum(1 for h in hist if h[0] == 0), ticks))
b.close()
