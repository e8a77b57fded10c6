"""AddressSanitizer + UndefinedBehaviorSanitizer builds of the CPU-side native code -- the host library (content loaders, track
build, reset poses: libpdbhost) and the test oracle (cpu_ref + rb) -- driven through their C entry points in a child process:
loaders on the synthetic content and on malformed blobs, 700 oracle ticks on the walled strip with body contacts and their response,
teleports by every mode, the auto-teleport hook.  (GPU sanitizers are not available on this pool; this is the CPU half.)"""
import os, subprocess, sys, tempfile, textwrap
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-O1', '-g']

DRIVER = textwrap.dedent('''
    import ctypes as C, os, sys, tempfile
    sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(tests)r)
    import numpy as np
    import pdb_ctypes as pc, synthetic_tracks
    host = C.CDLL(%(host)r); orc = C.CDLL(%(orc)r)
    host.pdb_last_error.restype = C.c_char_p
    host.pdb_teleport_to_spline.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    host.pdb_teleport_by_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    d = tempfile.mkdtemp(); synthetic_tracks.make_base(d, tracks=('flat', 'touge', 'walled'))
    synthetic_tracks.install_packed_car(d)
    blobs = {t: pc.build_track(host, d, t) for t in ('flat', 'touge', 'walled')}
    re = pc.build_track(host, d, 'touge', recompute_fat_points=True)
    assert len(re) == len(blobs['touge'])
    # malformed inputs must fail cleanly: a truncated surfaces.bin, an index past the vertex block, a missing track
    p = os.path.join(d, 'content', 'tracks', 'flat', 'surfaces.bin')
    raw = open(p, 'rb').read()
    for bad in (raw[:len(raw) // 2], raw[:-6] + b'\\xff\\xff' * 3):
        open(p, 'wb').write(bad)
        blob = C.c_void_p(); n = C.c_uint64()
        rc = host.pdb_build_track_opts(d.encode(), b'flat', 0, C.byref(blob), C.byref(n))
        assert rc != 0, 'malformed surfaces.bin accepted'
    open(p, 'wb').write(raw)
    blob = C.c_void_p(); n = C.c_uint64()
    assert host.pdb_build_track_opts(d.encode(), b'nosuchtrack', 0, C.byref(blob), C.byref(n)) != 0
    P = pc.CarParams()
    data = open(os.path.join(%(pkg)r, 'data', 'ks_toyota_ae86_drift.env.pdcar'), 'rb').read()
    C.memmove(C.byref(P), data, len(data))
    host.pdb_set_auto_teleport(C.byref(P), 1, 1, 2)
    trk = blobs['walled']
    S0 = pc.DynState(); assert host.pdb_initial_state(C.byref(P), trk, C.byref(S0)) == 0
    orc.cpuref_create.restype = C.c_void_p
    orc.cpuref_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    orc.cpuref_step_env.argtypes = [C.c_void_p, C.c_float, C.c_float]
    orc.cpuref_get_state.argtypes = [C.c_void_p, C.c_void_p]; orc.cpuref_set_state.argtypes = [C.c_void_p, C.c_void_p]
    orc.cpuref_set_auto_teleport_hook.argtypes = [C.c_void_p, C.c_void_p]; orc.cpuref_destroy.argtypes = [C.c_void_p]
    for b in range(P.numBodies):
        S0.body[b].pos[2] += 70.0; S0.body[b].lvel[2] = 14.0
    h = orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(S0))
    def tele(sp, mode):
        assert host.pdb_teleport_by_mode(C.byref(P), trk, mode, C.c_void_p(sp)) == 0
    hook = C.CFUNCTYPE(None, C.c_void_p, C.c_int)(tele)
    orc.cpuref_set_auto_teleport_hook(h, C.cast(hook, C.c_void_p))
    S = pc.DynState(); contacts = 0
    for t in range(700):
        orc.cpuref_step_env(h, 0.05, 1.0)
        orc.cpuref_get_state(h, C.byref(S)); contacts += S.numContacts > 0
        if t %% 200 == 199 and t > 450:
            for mode in (0, 1, 2):
                assert host.pdb_teleport_by_mode(C.byref(P), trk, mode, C.byref(S)) == 0
            orc.cpuref_set_state(h, C.byref(S))
    orc.cpuref_destroy(h)
    assert contacts > 0
    # round 6: pit boxes and location teleports (every box of the mountain road's pits.ini, ids outside the list, a point with nothing under it), and two oracle
    # cars exchanging their wakes (cpuref_get_slip / cpuref_set_other_slips)
    host.pdb_teleport_to_pit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    host.pdb_teleport_to_location.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    host.pdb_track_num_pits.argtypes = [C.c_void_p]; host.pdb_track_pit.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    trk = blobs['touge']
    np_ = host.pdb_track_num_pits(trk); assert np_ == 5
    S = pc.DynState(); assert host.pdb_initial_state(C.byref(P), trk, C.byref(S)) == 0
    m = (C.c_float * 16)()
    for pit in (-3, 0, 1, 2, 3, 4, 5, 1000):
        assert host.pdb_teleport_to_pit(C.byref(P), trk, pit, C.byref(S)) == 0
        assert (host.pdb_track_pit(trk, pit, m) == 0) == (0 <= pit < np_)
    for xyz in ((S.body[0].pos[0] + 1.0, S.body[0].pos[1] + 3.0, S.body[0].pos[2] - 2.0), (1.0e4, 50.0, -1.0e4)):
        assert host.pdb_teleport_to_location(C.byref(P), trk, C.c_float(xyz[0]), C.c_float(xyz[1]), C.c_float(xyz[2]), C.byref(S)) == 0
    trk = blobs['flat']
    Sa = pc.DynState(); Sb = pc.DynState()
    assert host.pdb_initial_state(C.byref(P), trk, C.byref(Sa)) == 0 and host.pdb_initial_state(C.byref(P), trk, C.byref(Sb)) == 0
    assert host.pdb_teleport_to_spline(C.byref(P), trk, C.c_float(0.0034), C.byref(Sb)) == 0
    for f in ('cpuref_get_slip', 'cpuref_set_slip'):
        getattr(orc, f).argtypes = [C.c_void_p, C.c_void_p]
    orc.cpuref_set_other_slips.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; orc.cpuref_set_guid.argtypes = [C.c_void_p, C.c_int]
    hs = [orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(Sa)), orc.cpuref_create(C.byref(P), trk, len(trk), C.byref(Sb))]
    orc.cpuref_set_guid(hs[1], 1)
    sl = (pc.SlipState * 2)()
    for t in range(600):
        for c in range(2):
            orc.cpuref_get_slip(hs[c], C.byref(sl[c]))
        for c in range(2):
            orc.cpuref_set_other_slips(hs[c], C.byref(sl[1 - c]), 1)
        orc.cpuref_step_env(hs[0], 0.0, 1.0); orc.cpuref_step_env(hs[1], 0.0, -0.5)
    for h in hs:
        orc.cpuref_destroy(h)
    print('sanitized run ok, contact ticks', contacts)
''')


@pytest.mark.skipif(not os.path.exists('/usr/bin/g++'), reason='needs g++')
def test_host_library_and_oracle_under_asan_ubsan():
    d = tempfile.mkdtemp(prefix='pdb_san_')
    csrc = os.path.join(ROOT, 'projectd-core_amd', 'csrc'); orc = os.path.join(ROOT, 'oracle'); inc = os.path.join(ROOT, 'include')
    host_so, orc_so = os.path.join(d, 'libpdbhost_san.so'), os.path.join(d, 'liboracle_san.so')
    base = ['g++', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared'] + SAN
    r = subprocess.run(base + ['-I' + inc, '-I' + os.path.join(csrc, 'host'), '-o', host_so] + [os.path.join(csrc, 'host', f) for f in ('model.cpp', 'track.cpp', 'reset.cpp', 'capi_host.cpp')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run(base + ['-mfma', '-I' + inc, '-I' + orc, '-o', orc_so] + [os.path.join(orc, 'cpu_ref', 'cpu_ref.cpp'), os.path.join(orc, 'cpu_ref', 'capi.cpp'), os.path.join(orc, 'rb', 'pdrb.cpp')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    asan = subprocess.run(['g++', '-print-file-name=libasan.so'], stdout=subprocess.PIPE, text=True).stdout.strip()
    script = os.path.join(d, 'driver.py')
    open(script, 'w').write(DRIVER % dict(pkg=os.path.join(ROOT, 'projectd-core_amd'), tests=os.path.join(ROOT, 'tests'), host=host_so, orc=orc_so))
    stdcpp = subprocess.run(['g++', '-print-file-name=libstdc++.so.6'], stdout=subprocess.PIPE, text=True).stdout.strip()
    # (libstdc++ next to libasan: the interceptor of __cxa_throw resolves the real one at start-up, before python has loaded any C++)
    env = dict(os.environ, LD_PRELOAD=asan + ' ' + stdcpp, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
    assert r.returncode == 0 and 'sanitized run ok' in r.stdout and 'AddressSanitizer' not in r.stdout and 'runtime error' not in r.stdout, r.stdout[-4000:]
