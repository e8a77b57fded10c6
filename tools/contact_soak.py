#!/usr/bin/env python3
"""GPU-only soak of the contact path (not part of the test suite): N cars with constant random actions and no resets on the playground-scale
mesh -- they end up leaning on, sitting in and scraping along obstacles -- for thousands of ticks; every 250 ticks every record is fetched and
checked for non-finite values, runaway positions and contact counts outside the row.  usage: contact_soak.py [cars=8192] [ticks=4000] [track=playground]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, parity_util as pu, pdbatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
track = sys.argv[3] if len(sys.argv) > 3 else 'playground'
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track(track)
b = pdbatch.Batch(n, P, trk, 0, 1)
b.set_seed(np.arange(1, n + 1, dtype=np.uint32) * 7919); b.reset(mode=2)
a = pu.make_actions(n, 4321); a[:, 0] *= 0.5
b.step_host(a, want_out=False)   # uploads the actions (and steps once)
worst = 0
for t0 in range(0, ticks, 250):
    b.step(250)
    st = b.get_state()
    raw = np.frombuffer(st, dtype=np.float32).reshape(n, -1)
    pos = np.array([[s.body[0].pos[0], s.body[0].pos[1], s.body[0].pos[2]] for s in st])
    nc = np.array([s.numContacts for s in st])
    bad = ~np.isfinite(pos).all(axis=1)
    far = np.abs(pos).max(axis=1) > 5000.0
    print('tick %5d: cars with live contact joints %5d (max %d per car), non-finite chassis %d, beyond 5 km %d, lowest chassis y %.2f' % (
        t0 + 250, int((nc > 0).sum()), int(nc.max()), int(bad.sum()), int(far.sum()), float(pos[:, 1].min())), flush=True)
    assert nc.min() >= 0 and nc.max() <= pc.MAX_CONTACTS
    worst = max(worst, int(bad.sum()) + int(far.sum()))
print('OK' if worst == 0 else 'FAILED: %d' % worst)
