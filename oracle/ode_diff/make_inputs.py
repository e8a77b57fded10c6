#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE.  Inputs for ode_diff: <out>/<car>.state.bin = the car's initial pdb_dyn_state on the synthetic flat
track (through the product's host library); the constant block is projectd-core_amd/data/<car>.env.pdcar as committed."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pdb_ctypes as pc, pdbatch
out = sys.argv[1] if len(sys.argv) > 1 else '.'
os.makedirs(out, exist_ok=True)
lib = pc.load_product(host_only=True)
import synthetic_tracks, tempfile
d = tempfile.mkdtemp(); synthetic_tracks.make_base(d, tracks=('flat',))
trk = pc.build_track(lib, d, 'flat')
for car in ('ks_toyota_ae86_drift', 'ks_toyota_supra_mkiv_drift', 'dthwsh_mazda_rx7_fc3s_sr20'):
    P = pdbatch.packed_params(car + '.env')
    S = pc.DynState()
    assert lib.pdb_initial_state(C.byref(P), trk, C.byref(S)) == 0
    open(os.path.join(out, car + '.state.bin'), 'wb').write(bytes(S))
    print(car, 'bodies', P.numBodies, 'joints', P.numJoints, 'rows', P.numRows)
