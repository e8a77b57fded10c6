"""Diagnostic (not a test): per-field single-tick GPU-vs-oracle deviation with state re-sync every tick."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, parity_util as pu, pdbatch, oracle_ctypes

def main(n=16, ticks=300, seed=7, resync=True, every=1):
    P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
    lib = pc.load_product(); orc = oracle_ctypes.load_oracle(portable_math=True)
    S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    b = pdbatch.Batch(n, P, trk, 0, 1)
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)) for _ in range(n)]
    acts = pu.make_actions(n, seed)
    maxabs = None; names = None; nexact = None; intbad = {}
    first = {}
    for t in range(ticks):
        if resync and t > 0:
            arr = (pc.DynState * n)()
            for i in range(n): orc.cpuref_get_state(hs[i], C.byref(arr[i]))
            b.set_state(arr)
        b.step_host(acts)
        for i in range(n): orc.cpuref_step_env(hs[i], float(acts[i, 0]), float(acts[i, 1]))
        sg = b.get_state()
        for i in range(n):
            sc = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(sc))
            fg, ig, fn, inn = pu.state_vectors(sg[i]); fc, ic, _, _ = pu.state_vectors(sc)
            d = np.abs(fg - fc); d[np.isnan(d)] = np.inf
            if maxabs is None: maxabs = np.zeros_like(d); names = fn; nexact = np.zeros(len(d), int); refmax = np.zeros_like(d)
            maxabs = np.maximum(maxabs, d); nexact += (fg == fc); refmax = np.maximum(refmax, np.abs(fc))
            for k in np.where(d > 0)[0]:
                if names[k] not in first: first[names[k]] = (t, i, fg[k], fc[k])
            for k in np.where(ig != ic)[0]: intbad.setdefault(inn[k], []).append((t, i, int(ig[k]), int(ic[k])))
    tot = ticks * n
    order = np.argsort(-maxabs / np.maximum(refmax, 1e-30))
    print('fields exact in all samples: %d of %d' % (int((nexact == tot).sum()), len(names)))
    print('%-32s %12s %12s %8s  first(t,car,gpu,cpu)' % ('field', 'maxabs', 'refmax', 'exact%'))
    for k in order[:60]:
        if maxabs[k] == 0: break
        print('%-32s %12.4e %12.4e %7.1f%%  %s' % (names[k], maxabs[k], refmax[k], 100.0 * nexact[k] / tot, first.get(names[k])))
    print('int mismatches:', {k: v[:3] for k, v in intbad.items()})
    b.close()

if __name__ == '__main__':
    if len(sys.argv) > 3:
        main(int(sys.argv[1]), int(sys.argv[2]), 7, sys.argv[3] == 'resync')
    elif len(sys.argv) > 2:
        main(int(sys.argv[1]), int(sys.argv[2]))
    else:
        main()
