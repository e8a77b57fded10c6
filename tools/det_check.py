#!/usr/bin/env python3
"""Diagnostic (GPU box): N cars on the playground in env mode for some ticks, partitioned like the bench; prints a checksum of every record and the
contact passes' load.  Run it with different PDB_LIB / PDB_CONTACT_GRID settings: the checksum must not change."""
import os, sys, hashlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, parity_util as pu, pdbatch, projectd_env as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 320
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 3
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('playground')
b = pdbatch.Batch(n, P, trk, 0, 1)
b.set_seed(np.arange(1, n + 1, dtype=np.uint32) * 7919); b.reset(mode=2)
a = pu.make_actions(n, 4321); a[:, 0] *= 0.5
b.step_host(a, want_out=False)
b.set_env(E.EnvConfig())
if parts > 1:
    b.set_partitions(parts)
    b.step_ring(ticks)
else:
    b.step(ticks)
b.sync()
st = b.get_state()
print('lib %s grid %s parts %d: %s  contact-pass load %s' % (os.environ.get('PDB_LIB', 'in-tree'), os.environ.get('PDB_CONTACT_GRID', 'adaptive'), parts,
      hashlib.sha1(bytes(st)).hexdigest()[:16], [b.lib.pdb_contact_pass_load(b.h, q) for q in range(parts if parts > 1 else 1)]))
