"""Diagnostic: growth of the free-running GPU-vs-oracle deviation (group-normalised metric) over time."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, parity_util as pu, pdbatch, oracle_ctypes
n, ticks = int(sys.argv[1]), int(sys.argv[2])
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
lib = pc.load_product(); orc = oracle_ctypes.load_oracle(portable_math=True)
S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
b = pdbatch.Batch(n, P, trk, 0, 1)
hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n)]
acts = pu.make_actions(n, 1234)
intbad = 0
for t in range(ticks):
    b.step_host(acts)
    for i in range(n): orc.cpuref_step_env(hs[i], float(acts[i, 0]), float(acts[i, 1]))
    if t % 50 == 49 or t == ticks - 1:
        sg = b.get_state(); w = 0; wi = None; nb = 0
        for i in range(n):
            sc = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(sc))
            rel, name, vg, vc, bad = pu.compare_states(sg[i], sc)
            nb += 1 if bad else 0
            if rel > w: w = rel; wi = (i, name, vg, vc)
        print('tick %5d worst rel %.3e  cars with int mismatch %d/%d  %s' % (t, w, nb, n, wi))
b.close()
