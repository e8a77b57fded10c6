"""Diagnostic: per-phase shader-clock shares of pdb_step_kernel (libpdbatch_stamps.so, -DPDB_STAMPS)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, parity_util as pu, pdbatch
pc_load = pc.load_product
def load_stamps():
    lib = pc._load(os.path.join(pc.PKG, 'libpdbatch_stamps.so'))
    return lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
orig = pc.load_product
def patched(host_only=False):
    if host_only: return orig(True)
    lib = orig.__wrapped__() if hasattr(orig, '__wrapped__') else None
    return lib
# build a Batch on the stamps library
lib = C.CDLL(os.path.join(pc.PKG, 'libpdbatch_stamps.so'))
import types
def lp(host_only=False):
    l = lib
    l.pdb_last_error.restype = C.c_char_p
    l.pdb_create.restype = C.c_void_p; l.pdb_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    l.pdb_destroy.argtypes = [C.c_void_p]; l.pdb_step_host.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    l.pdb_step_n.argtypes = [C.c_void_p, C.c_float, C.c_int]; l.pdb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    l.pdb_get_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]; l.pdb_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return l
pc.load_product = lp
P = pdbatch.packed_params()
pc.load_product = orig
trk = pdbatch.synthetic_track('flat')
pc.load_product = lp
b = pdbatch.Batch(n, P, trk, 0, 1)
a = pu.make_actions(n, 1234)
for _ in range(400): b.step_host(a)
st = np.zeros((n, 16), dtype=np.uint64)
lib.pdb_debug_stamps(b.h, st.ctypes.data_as(C.c_void_p))
sti = st.astype(np.int64)
print('solveRows detail (median cycles): setup(JM,rhs) %d | A rows %d | factorisation %d | fwd+bwd %d' % (np.median(sti[:,14]-sti[:,7]), np.median(sti[:,8]-sti[:,14]), np.median(sti[:,15]-sti[:,8]), np.median(sti[:,9]-sti[:,15])))
sti[:, 8] = sti[:, 9]
d = np.diff(sti[:, :14], axis=1)
names = ['load', 'phase1 lane0 pre-step', 'susp (lane=wheel)', 'tyre (lane=wheel)', 'wings/steer/assists/drivetrain/ARB', 'accumulate', 'bodies+joint rows+JinvM+rhs', 'A assembly', 'LDLT', 'substitution', 'cforce+integrate', 'postStep track+scoring', 'outputs+store']
tot = d.sum(1)
print('cars %d: median wave lifetime %.0f shader clocks (= %.1f us at 100 MHz memtime?)' % (n, np.median(tot), np.median(tot) / 100.0))
for i, nm in enumerate(names):
    print('%-40s median %8.0f  share %5.1f%%' % (nm, np.median(d[:, i]), 100.0 * np.median(d[:, i]) / np.median(tot)))
b.close()
