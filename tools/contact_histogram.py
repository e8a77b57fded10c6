#!/usr/bin/env python3
"""Contact-point candidates per car and odd frame against the PDB_MAX_CONTACTS cut: N cars with constant random
actions on a synthetic track, stepped by the CPU oracle (diagnostic: test infrastructure, not product).
usage: contact_histogram.py [track=playground] [cars=32] [ticks=2500] [model]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, oracle_ctypes, pdbatch, parity_util

track = sys.argv[1] if len(sys.argv) > 1 else 'playground'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
model = sys.argv[4] if len(sys.argv) > 4 else 'ks_toyota_ae86_drift'
P = pdbatch.packed_params(model + '.env')
trk = pdbatch.synthetic_track(track)
lib = pc.load_product(host_only=True); orc = oracle_ctypes.load_oracle(portable_math=True)
orc.cpuref_contact_candidates.argtypes = [C.c_void_p]
S0 = pc.DynState()
assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n)]
acts = parity_util.make_actions(n, 7)
hist = np.zeros(4096, dtype=np.int64)
kinds = np.zeros(2, dtype=np.int64)
S = pc.DynState()
MAXR = 3 * pc.MAX_CONTACTS
lam = (C.c_float * MAXR)(); lo = (C.c_float * MAXR)(); hi = (C.c_float * MAXR)(); itn = C.c_int()
rounds = []
for t in range(ticks):
    for i, h in enumerate(hs):
        orc.cpuref_step_env(h, float(acts[i, 0]), float(acts[i, 1]))
        orc.cpuref_get_state(h, C.byref(S))
        if S.simFrame % 2 == 0:   # the frame just collided was odd
            c = orc.cpuref_contact_candidates(h)
            hist[min(c, 4095)] += 1
        if S.numContacts > 0:
            orc.cpuref_last_contact_rows(h, lam, lo, hi, MAXR, C.byref(itn)); rounds.append(itn.value)
tot = hist[1:].sum()
print('track %s, %d cars x %d ticks (%s): %d odd frames, %d with contact points' % (track, n, ticks, model, hist.sum(), tot))
edges = [1, 2, 4, 7, 11, 17, 22, 33, 65, 129, 4096]
for a, b in zip(edges[:-1], edges[1:]):
    print('  %4d..%-4d candidates: %7d  (%.1f %% of frames in contact)' % (a, b - 1, hist[a:b].sum(), 100.0 * hist[a:b].sum() / max(tot, 1)))
print('  max candidates: %d; frames above 10: %.1f %%, above 21: %.1f %%, above 32: %.1f %%' % (
    np.nonzero(hist)[0].max(), 100.0 * hist[11:].sum() / max(tot, 1), 100.0 * hist[22:].sum() / max(tot, 1), 100.0 * hist[33:].sum() / max(tot, 1)))
if rounds:
    r = np.sort(np.array(rounds))
    print('  LCP rounds per solve (both stages, %d solves): median %d, p90 %d, p95 %d, p99 %d, max %d' % (len(r), r[len(r) // 2], r[int(len(r) * 0.9)], r[int(len(r) * 0.95)], r[int(len(r) * 0.99)], r[-1]))
