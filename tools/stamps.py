"""Diagnostic: per-phase shader-clock shares of pdb_step_kernel (libpdbatch_stamps.so, -DPDB_STAMPS).  One tick per launch with a host
synchronisation in between: the device idles between launches, and phases that wait for memory read longer here than under the bench's
continuous load (a round trip measured 9 us in this tool) -- read the compute phases and the shares, not the loads."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc, parity_util as pu, pdbatch
pc_load = pc.load_product
def load_stamps():
    lib = pc._load(os.path.join(pc.ROOT, 'tools', 'variants', os.environ.get('PDB_STAMPS_LIB', 'libpdbatch_stamps.so')))
    return lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
orig = pc.load_product
def patched(host_only=False):
    if host_only: return orig(True)
    lib = orig.__wrapped__() if hasattr(orig, '__wrapped__') else None
    return lib
# build a Batch on the stamps library
lib = C.CDLL(os.path.join(pc.ROOT, 'tools', 'variants', os.environ.get('PDB_STAMPS_LIB', 'libpdbatch_stamps.so')))
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
kind = sys.argv[2] if len(sys.argv) > 2 else 'flat'
gen = {'step': float(sys.argv[3])} if len(sys.argv) > 3 else {}
trk = pdbatch.reference_track(kind) if kind in pdbatch.REFERENCE_TRACKS else pdbatch.synthetic_track(kind, **gen)
pc.load_product = lp
b = pdbatch.Batch(n, P, trk, 0, 1)
a = pu.make_actions(n, 1234)
if kind == 'touge':   # spread the cars around the lap, drive them with a mild constant action
    st = b.get_state()
    hl = pc.load_product(host_only=True)
    for i in range(n):
        hl.pdb_teleport_to_spline(C.byref(P), trk, C.c_float((i % 4096) / 4096.0), C.byref(st[i]))
    b.set_state(st)
    a[:, 0] = 0.0; a[:, 1] = -0.5
if kind in ('playground', 'nordring') or kind in pdbatch.REFERENCE_TRACKS:   # every car to its own random point of the lap (device teleports), driven by a mild constant action
    lib.pdb_set_seed.argtypes = [C.c_void_p, C.c_void_p]; lib.pdb_reset_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    b.set_seed(np.arange(1, n + 1, dtype=np.uint32) * 7919)
    b.reset(mode=2)
    a[:, 0] *= 0.3; a[:, 1] = -0.5
for _ in range(int(os.environ.get('PDB_STAMP_TICKS', '400'))): b.step_host(a)
st = np.zeros((2 * n, 32), dtype=np.uint64)
lib.pdb_debug_stamps(b.h, st.ctypes.data_as(C.c_void_p))
full = st[n:].astype(np.int64)     # the contact pass's stamps of the last tick (cars it held)
st = st[:n]
sti = st.astype(np.int64)
CPB = 3
first = sti[::CPB]                      # first car of each block also carries the pack wave's stamps (4, 14, 15)
med = lambda x: float(np.median(x))
t0 = sti[:, 0]
print('cars %d (car wave, median shader clocks since wave start):' % n)
for nm, k in (('record loaded, ERP set', 1), ('world inertia + non-steer joint rows done (barrier 1)', 2), ('steer rows done, A assembly starts', 7),
              ('barrier 2 passed (forces ready)', 5), ('late bodies done, rhs starts', 6), ('lambda solved', 9),
              ('integration done', 11), ('post scans done', 14), ('pack scoring done (barrier)', 12), ('record stored', 13)):
    print('  %-58s %8.0f' % (nm, med(sti[:, k] - t0)))
print('pack wave (median, relative to the block\'s first car wave start):')
for nm, k in (('pre-step + steering rods done', 4), ('tyres done', 24), ('suspensions done', 30), ('drivetrain done', 15), ('tyre tail (thermal) done', 29), ('post: barrier passed', 25), ('post: locator done', 26), ('post: look-ahead done', 27), ('post: scoring done', 28)):
    print('  %-58s %8.0f' % (nm, med(first[:, k] - first[:, 0])))
print('pack internals (car 0 of the block, wheel 0): pre-step end -> hub matrix %d | ray cast %d | contact + SCTM + forces %d | torque/lock %d | thermal %d ;; drive: tyres end -> before drivetrainStep %d | drivetrainStep %d' % (med(first[:,16]-first[:,4]), med(first[:,17]-first[:,16]), med(first[:,18]-first[:,17]), med(first[:,19]-first[:,18]), med(first[:,20]-first[:,19]), med(first[:,22]-first[:,24]), med(first[:,23]-first[:,22])))
print('pack internals, 99th percentile / max: hub matrix %d/%d | ray cast %d/%d | contact + SCTM + forces %d/%d ;; tyres done %d/%d, drivetrain done %d/%d' % (
    np.percentile(first[:,16]-first[:,4], 99), (first[:,16]-first[:,4]).max(), np.percentile(first[:,17]-first[:,16], 99), (first[:,17]-first[:,16]).max(), np.percentile(first[:,18]-first[:,17], 99), (first[:,18]-first[:,17]).max(),
    np.percentile(first[:,24]-first[:,0], 99), (first[:,24]-first[:,0]).max(), np.percentile(first[:,15]-first[:,0], 99), (first[:,15]-first[:,0]).max()))
print('wave lifetime median %.0f, 99th percentile %.0f, max %.0f clocks' % (med(sti[:, 13] - t0), np.percentile(sti[:, 13] - t0, 99), (sti[:, 13] - t0).max()))
b.close()

held = (full[:, 0] != 0) & (full[:, 13] > full[:, 13].max() - 1000000)   # rows of the last tick's pass (a car that left the pass keeps its old row; a tick here is over a million clocks)
if held.any():
    f = full[held]
    print('contact pass: %d of %d cars in it on the last tick; median / 90th percentile / max shader clocks since the wave started the car:' % (held.sum(), n))
    for nm, k in (('snapshot loads issued', 7), ('snapshot in LDS', 8), ('record loaded (workgroup barrier)', 1), ('joint rows done, collision pass starts', 16), ('collision pass done', 17), ('forces ready (barrier 2)', 5),
                  ('lambda of the unbounded rows', 9), ('integration done', 11), ('post scans done', 14), ('record stored', 13)):
        d = f[:, k] - f[:, 0]
        d = d[f[:, k] != 0]
        if len(d): print('  %-48s %9.0f %9.0f %9.0f' % (nm, np.median(d), np.percentile(d, 90), d.max()))
    cs = f[f[:, 19] != 0]
    if len(cs): print('  contact solve (cars with live joints: %d): median %.0f, max %.0f clocks' % (len(cs), np.median(cs[:, 19] - cs[:, 18]), (cs[:, 19] - cs[:, 18]).max()))
    pk = f[(f[:, 4] > f[:, 0]) & (f[:, 28] > f[:, 4]) & (f[:, 28] - f[:, 0] < 20000000) & (f[:, 24] > f[:, 4]) & (f[:, 15] > f[:, 24])]
    if len(pk):
        print('  pack wave of the contact pass (groups: %d), since the group\'s first car wave started:' % len(pk))
        for nm, k in (('pre-step + steering rods done', 4), ('tyres done', 24), ('suspensions done', 30), ('drivetrain done', 15), ('tyre tail (thermal) done', 29), ('post: barrier passed', 25), ('post: scoring done', 28)):
            d = pk[:, k] - pk[:, 0]
            print('    %-44s %9.0f %9.0f %9.0f' % (nm, np.median(d), np.percentile(d, 90), d.max()))
    cn = f[f[:, 20] != 0]
    if len(cn):
        u = cn.astype(np.uint64)
        cols = {'turns': u[:, 20] >> 32, 'broad-phase survivors': u[:, 20] & 0xffffffff, 'dense turns': u[:, 21] >> 32, 'wall triangles one by one': u[:, 21] & 0xffffffff,
                'road triangles (box)': u[:, 22] >> 32, 'pairs after the plane-side rejections': u[:, 22] & 0xffffffff, 'contact candidates': u[:, 23]}
        coll = (cn[:, 17] - cn[:, 16]).astype(np.float64)
        print('  collision pass counters (%d cars): median / 90th percentile / max, and the correlation with the pass\'s clocks' % len(cn))
        for nm, v in cols.items():
            v = v.astype(np.float64)
            cc = np.corrcoef(v, coll)[0, 1] if v.std() > 0 else 0.0
            print('    %-40s %8.0f %8.0f %8.0f   r = %.2f' % (nm, np.median(v), np.percentile(v, 90), v.max(), cc))
        A = np.stack([np.ones(len(cn))] + [v.astype(np.float64) for v in cols.values()], axis=1)
        coef = np.linalg.lstsq(A, coll, rcond=None)[0]
        print('    least squares, clocks ~ ' + ' + '.join(['%.0f' % coef[0]] + ['%.0f x %s' % (c, nm) for c, nm in zip(coef[1:], cols.keys())]))
    t00 = f[:, 0].min()
    print('  pass timeline (clocks since its first car started): car starts median %.0f, 90th %.0f, max %.0f; records stored median %.0f, max %.0f' % (
        np.median(f[:, 0] - t00), np.percentile(f[:, 0] - t00, 90), (f[:, 0] - t00).max(), np.median(f[:, 13] - t00), (f[:, 13] - t00).max()))
