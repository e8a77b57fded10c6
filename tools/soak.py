#!/usr/bin/env python3
"""Long free-running GPU-vs-oracle soak (not part of the test suite): every car model, flat plane with random constant actions
and the mountain road with the feedback controller, tens of thousands of ticks, bit-exact state comparison every 100 ticks."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import numpy as np
try:
    import torch; torch.cuda.is_available() and torch.cuda.init()
except Exception:
    pass
import pdb_ctypes as pc, parity_util as pu, pdbatch, oracle_ctypes, sharding
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = 32
lib = pc.load_product(); orc = oracle_ctypes.load_oracle(True)
for model in ('ks_toyota_ae86_drift', 'ks_toyota_supra_mkiv_drift', 'gravygarage_street_ae86_readie', 'pdb_heave_rx7'):
    for track in ('flat', 'touge'):
        P = pdbatch.packed_params(model + '.env'); trk = pdbatch.synthetic_track(track)
        S0 = pc.DynState(); assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
        starts = (pc.DynState * n)()
        for i in range(n):
            s = pc.DynState.from_buffer_copy(bytes(S0))
            if track == 'touge':
                lib.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(i / n), C.byref(s))
            C.memmove(C.byref(starts[i]), C.byref(s), C.sizeof(s))
        b = pdbatch.Batch(n, P, trk, device=0, action_mode=1); b.set_state(starts)
        hs = []
        for i in range(n):
            h = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0)); orc.cpuref_set_state(h, C.byref(starts[i])); hs.append(h)
        a = sharding.global_actions(n, 77) if track == 'flat' else np.zeros((n, 2), np.float32)
        t0 = time.time(); bad = None
        for t in range(ticks):
            out = b.step_host(a)
            for i in range(n):
                orc.cpuref_step_env(hs[i], float(a[i, 0]), float(a[i, 1]))
            if track == 'touge':
                obs = np.ascontiguousarray(out['obs'], dtype=np.float32)
                for i in range(n):
                    orc.cpuref_scenario_feedback(6, t, obs[i].ctypes.data_as(C.c_void_p), a[i].ctypes.data_as(C.c_void_p))
                # cars that left the road or got stuck restart at their slot (both sides)
                for i in np.where(out['flags'] != 0)[0]:
                    s = pc.DynState.from_buffer_copy(bytes(starts[i]))
                    arr = (pc.DynState * 1)(); C.memmove(C.byref(arr[0]), C.byref(s), C.sizeof(s))
                    b.set_state(arr, first=int(i)); orc.cpuref_set_state(hs[i], C.byref(s)); a[i] = 0
            if t % 100 == 99:
                sg = b.get_state()
                for i in range(n):
                    sc = pc.DynState(); orc.cpuref_get_state(hs[i], C.byref(sc))
                    rel, name, vg, vc, bi = pu.compare_states(sg[i], sc)
                    if bi or rel != 0.0:
                        bad = (t, i, name, vg, vc, bi[:3]); break
                if bad:
                    break
        print('%-34s %-6s %6d ticks x %d cars: %s  (%.0f s)' % (model, track, ticks, n, 'BIT-EXACT' if not bad else 'MISMATCH %s' % (bad,), time.time() - t0), flush=True)
        b.close()
        for h in hs:
            orc.cpuref_destroy(h)
