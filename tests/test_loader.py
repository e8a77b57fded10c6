"""Host logic behind the boundary: car-model / track builders, tunes, scoring vars, reset poses (csrc/host/*).
Tests that need the reference's shipped car data run only where /root/reference exists (the build container);
the packed AE86 block they check (projectd-core_amd/data/*.pdcar) is what travels to the GPU box."""
import ctypes as C, os, struct, sys
import numpy as np
import pytest
import pdb_ctypes as pc
from conftest import car_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'content', 'cars')), reason='reference content not present')
AE86 = 'ks_toyota_ae86_drift'


def header(blob):
    return pc.TrackHeader.from_buffer_copy(blob[:C.sizeof(pc.TrackHeader)]) if hasattr(pc, 'TrackHeader') else None


@needs_ref
def test_packed_ae86_block_is_reproducible(hostlib):
    """the committed .pdcar blocks are exactly what the loader produces from the shipped car data + env settings"""
    P = pc.env_params(hostlib, REF)
    packed = open(os.path.join(ROOT, 'projectd-core_amd', 'data', AE86 + '.env.pdcar'), 'rb').read()
    assert bytes(P) == packed
    D = pc.CarParams()
    assert hostlib.pdb_build_car_model(REF.encode(), AE86.encode(), C.byref(D)) == 0
    assert bytes(D) == open(os.path.join(ROOT, 'projectd-core_amd', 'data', AE86 + '.default.pdcar'), 'rb').read()


@needs_ref
def test_tuned_scenario_blocks_are_reproducible(hostlib, oracle):
    """<scenario>.tuned.pdcar (what the tunes* scenarios start from on a machine without the cars' setup.ini) = the env block +
    the scenario's setCarTune list through pdb_set_car_tune, exactly as tests/scenario_util.setup applies it here"""
    seen = 0
    for sid in range(oracle.cpuref_num_scenarios()):
        nm = C.c_char_p(); val = C.c_float()
        if not oracle.cpuref_scenario_tune(sid, 0, C.byref(nm), C.byref(val)):
            continue
        name = oracle.cpuref_scenario_name(sid).decode(); model = oracle.cpuref_scenario_car(sid).decode()
        P = pc.env_params(hostlib, REF, model)
        i = 0
        while oracle.cpuref_scenario_tune(sid, i, C.byref(nm), C.byref(val)):
            hostlib.pdb_set_car_tune(C.byref(P), REF.encode(), model.encode(), nm.value, val.value, 0)
            i += 1
        assert bytes(P) == open(os.path.join(ROOT, 'projectd-core_amd', 'data', name + '.tuned.pdcar'), 'rb').read(), name
        seen += 1
    assert seen == 3


@needs_ref
def test_track_packs_hold_the_blobs_the_loader_builds(hostlib, built):
    """projectd-core_amd/data/tracks/*.pdtrack.z (git-ignored build output of __graft_entry__.build(), tools/pack_tracks.py): the
    reference's six tracks as the product's blob, for the GPU box; each pack decompresses to what pdb_build_track gives here"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pack_tracks, tempfile, synthetic_tracks
    scratch = synthetic_tracks.make_base(tempfile.mkdtemp(prefix='pdb_packtest_'), tracks=())
    for name in pack_tracks.MESH_TRACKS + pack_tracks.RIBBON_TRACKS + pack_tracks.WALLED_RIBBONS:
        assert pc.load_track_pack(name) == pack_tracks.build_blob(hostlib, name, scratch), name


@needs_ref
def test_scenarios_set_up_identically_without_the_reference_content(hostlib, oracle, base_dir, monkeypatch):
    """what the GPU box sees (no /root/reference: packed tracks, packed tuned blocks) is byte for byte what the build container sees"""
    import scenario_util as SU
    for sid in range(oracle.cpuref_num_scenarios()):
        monkeypatch.setattr(SU, 'REF_CONTENT', REF + '/content')
        here = SU.setup(oracle, hostlib, sid, base_dir)
        monkeypatch.setattr(SU, 'REF_CONTENT', '/nonexistent/content')
        there = SU.setup(oracle, hostlib, sid, base_dir)
        assert bytes(here['P']) == bytes(there['P']) and here['blob'] == there['blob'] and bytes(here['S0']) == bytes(there['S0']), here['name']


@needs_ref
def test_lane_tune_rows_cover_what_the_env_tunes_and_scoring_vars_write(hostlib):
    """pdb_lane_tune (the per-lane override row) holds exactly the fields of the car block that projectd_env.py's eight setCarTune calls and its
    setScoringVar calls change: a block tuned through pdb_set_car_tune / pdb_set_scoring_var differs from the untuned one ONLY inside those fields,
    and the row read out of it carries the new values"""
    base = pc.env_params(hostlib, REF)
    tuned = pc.CarParams.from_buffer_copy(bytes(base))
    vals = {'FRONT_BIAS': 62.0, 'DIFF_POWER': 45.0, 'DIFF_COAST': 20.0, 'FINAL_RATIO': 4.3, 'PRESSURE_LF': 31.0, 'PRESSURE_RF': 30.0, 'PRESSURE_LR': 26.0, 'PRESSURE_RR': 25.0}
    assert set(vals) == set(pc.ENV_TUNES)
    for k, v in vals.items():
        assert hostlib.pdb_set_car_tune(C.byref(tuned), REF.encode(), AE86.encode(), k.encode(), v, 0) == 0
    for j, k in enumerate(pc.ENV_SCORING):
        assert hostlib.pdb_set_scoring_var(C.byref(tuned), k.encode(), 0.25 + j) == 0
    covered = set()
    for name, size in (('finalRatio', 8), ('diffPowerRamp', 8), ('diffCoastRamp', 8), ('frontBias', 4), ('scoring', C.sizeof(pc.Scoring))):
        off = getattr(pc.CarParams, name).offset
        covered.update(range(off, off + size))
    for w in range(4):
        off = pc.CarParams.tyre.offset + w * C.sizeof(pc.Tyre) + pc.Tyre.pressureStatic.offset
        covered.update(range(off, off + 4))
    a, b = bytes(base), bytes(tuned)
    diff = {i for i in range(len(a)) if a[i] != b[i]}
    assert diff and diff <= covered, sorted(diff - covered)[:8]
    row = pc.LaneTune()
    assert hostlib.pdb_lane_tune_from_params(C.byref(tuned), C.byref(row)) == 0
    assert row.valid == 1 and row.finalRatio == tuned.finalRatio == float(np.float32(4.3)) and abs(row.frontBias - 0.62) < 1e-6
    assert [row.pressureStatic[w] for w in range(4)] == [31.0, 30.0, 26.0, 25.0]
    assert bytes(row.scoring) == bytes(tuned.scoring) and row.scoring.StallPenalty == 0.25 + len(pc.ENV_SCORING) - 1


@needs_ref
def test_ae86_topology(hostlib):
    """7 bodies, 16 joints, 33 constraint rows; RWD, 5 forward gears + R + N"""
    P = pc.CarParams()
    assert hostlib.pdb_build_car_model(REF.encode(), AE86.encode(), C.byref(P)) == 0
    assert (P.numBodies, P.numJoints, P.numRows) == (7, 16, 33)
    rows = {0: 6, 1: 3, 2: 5, 3: 1}   # fixed, ball, slider, dball (pdb_types.h joint types)
    assert sum(rows[P.joints[j].type] for j in range(P.numJoints)) == 33
    total = sum(P.bodies[b].mass for b in range(P.numBodies))
    assert 900.0 < total < 1400.0
    assert P.numGears >= 6 and P.gearRatio[1] == 0.0      # [R, N, 1..]
    assert P.powerCurve.n > 4 and P.numWings >= 1
    assert abs(P.gravity[1] + 9.80665) < 1e-3 or abs(P.gravity[1] + 9.81) < 0.05


@needs_ref
def test_tunes_follow_setup_spinner_semantics(hostlib):
    """projectd_env.py:127-130 tunes -> SetupManager::setTune (Car/SetupManager.cpp): FINAL_RATIO is a ratio index-free
    value, FRONT_BIAS a percentage, DIFF_* percentages, PRESSURE_* psi"""
    P = pc.CarParams()
    assert hostlib.pdb_build_car_model(REF.encode(), AE86.encode(), C.byref(P)) == 0
    before = (P.finalRatio, P.frontBias, P.diffPowerRamp, P.tyre[0].pressureStatic)
    for k, v in pc.ENV_TUNES.items():
        assert hostlib.pdb_set_car_tune(C.byref(P), REF.encode(), AE86.encode(), k.encode(), v, 0) == 0
    assert P.finalRatio == 5.0
    assert abs(P.frontBias - 0.55) < 1e-6
    assert abs(P.diffPowerRamp - 0.30) < 1e-6 and abs(P.diffCoastRamp - 0.30) < 1e-6
    assert all(abs(P.tyre[i].pressureStatic - 28.0) < 1e-6 for i in range(4))
    assert before != (P.finalRatio, P.frontBias, P.diffPowerRamp, P.tyre[0].pressureStatic)
    # unknown tune names are ignored, like the reference (SetupManager::setTune looks the name up and returns)
    snap = bytes(P)
    assert hostlib.pdb_set_car_tune(C.byref(P), REF.encode(), AE86.encode(), b'NO_SUCH_TUNE', 1.0, 0) == 0
    assert bytes(P) == snap


@needs_ref
def test_unsupported_or_missing_models_fail_with_a_message(hostlib):
    P = pc.CarParams()
    assert hostlib.pdb_build_car_model(REF.encode(), b'no_such_car', C.byref(P)) < 0
    assert hostlib.pdb_last_error()
    ok = 0
    for m in sorted(os.listdir(os.path.join(REF, 'content', 'cars'))):
        rc = hostlib.pdb_build_car_model(REF.encode(), m.encode(), C.byref(P))
        if rc == 0:
            ok += 1
            assert P.numRows <= 40 and P.numBodies <= 8
        else:   # e.g. double-wishbone / turbo cars: refused loudly, never half-loaded
            assert len(hostlib.pdb_last_error()) > 0
    assert ok >= 1


def test_scoring_vars_roundtrip(hostlib, env_params):
    P = pc.CarParams.from_buffer_copy(bytes(env_params))
    for k, v in pc.ENV_SCORING.items():
        assert hostlib.pdb_get_scoring_var(C.byref(P), k.encode()) == np.float32(v)
    assert hostlib.pdb_set_scoring_var(C.byref(P), b'DriftBonus', 2.5) == 0
    assert hostlib.pdb_get_scoring_var(C.byref(P), b'DriftBonus') == 2.5
    assert hostlib.pdb_set_scoring_var(C.byref(P), b'NoSuchVar', 1.0) < 0
    assert b'NoSuchVar' in hostlib.pdb_last_error()


def test_assists_switches(hostlib, env_params):
    P = pc.CarParams.from_buffer_copy(bytes(env_params))
    assert (P.acUseOnStart, P.acUseOnChange, P.autoShiftActive, P.autoBlipActive, P.smoothSteer) == (1, 1, 1, 1, 1)
    hostlib.pdb_set_assists(C.byref(P), 0, 1, 0, 0)
    assert (P.acUseOnStart, P.acUseOnChange, P.autoShiftActive, P.autoBlipActive, P.smoothSteer) == (0, 0, 1, 0, 0)


def test_flat_track_blob(hostlib, flat_track):
    """synthetic config-1/2 track: 1 surface, 2 triangles, 301 slim points -> fat points, B-spline nodes"""
    magic, version, nsurf, ntris, nfat, nnodes, istep, closed = struct.unpack_from('<8i', flat_track, 0)
    tlen, twidth, grip, cell = struct.unpack_from('<4f', flat_track, 32)
    offs = struct.unpack_from('<7Q', flat_track, 48)
    assert magic == int.from_bytes(b'PDTK', 'little') or magic == int.from_bytes(b'KTDP', 'little')
    assert (nsurf, ntris, nfat, closed) == (1, 2, 301, 0)
    assert nnodes > nfat
    assert abs(tlen - 3000.0) < 1.0 and abs(twidth - 12.0) < 1e-3
    assert offs[-1] == len(flat_track)
    fat = np.frombuffer(flat_track, dtype='<f4', count=nfat * 15, offset=offs[2]).reshape(nfat, 15)
    assert np.allclose(fat[:, 0], 0) and np.allclose(np.diff(fat[:, 2]), 10.0)          # best line along +z every 10 m
    assert np.allclose(np.abs(fat[:, 3] - fat[:, 6]), 12.0, atol=1e-3)                  # left/right 6 m each side
    assert np.allclose(fat[:-1, 12:15], [0, 0, 1], atol=1e-6)                           # forwardDir
    dist = np.frombuffer(flat_track, dtype='<f4', count=nfat, offset=offs[3])
    assert dist[0] == np.float32(0.1) and np.all(np.diff(dist) > 0)   # Track::initTrackPoints starts the running length at 0.1 (Track.cpp:200-271)


def test_missing_track_fails_with_a_message(hostlib, base_dir):
    blob = C.c_void_p(); n = C.c_uint64()
    assert hostlib.pdb_build_track(base_dir.encode(), b'no_such_track', C.byref(blob), C.byref(n)) < 0
    assert len(hostlib.pdb_last_error()) > 0
    assert hostlib.pdb_build_track(None, b'flat', C.byref(blob), C.byref(n)) == -1


@needs_ref
def test_driftplayground_mesh_counts(hostlib):
    """510 surfaces (490 WALL + 20 TRACK), 112 411 triangles"""
    import tempfile, shutil, synthetic_tracks
    d = tempfile.mkdtemp(prefix='pdb_dp_')
    try:
        synthetic_tracks.make_base(d, tracks=())
        os.makedirs(os.path.join(d, 'content', 'tracks'), exist_ok=True)
        shutil.copytree(os.path.join(REF, 'content', 'tracks', 'driftplayground'), os.path.join(d, 'content', 'tracks', 'driftplayground'))
        os.system('chmod -R u+w "%s"' % d)
        blob = pc.build_track(hostlib, d, 'driftplayground')
    finally:
        shutil.rmtree(d, ignore_errors=True)
    magic, version, nsurf, ntris, nfat, nnodes = struct.unpack_from('<6i', blob, 0)
    assert (nsurf, ntris) == (510, 112411)
    assert nfat > 100


def test_initial_state_and_spline_teleport(hostlib, env_params, flat_track, state0):
    """teleportCarByMode(Start) with no pits falls back to spline start; teleportToSpline(d) puts the chassis on the
    best line facing forwardDir (Car::teleportToSpline -> forceRotation + forcePosition, Car.cpp)"""
    s = state0
    ch = s.body[0]
    assert abs(ch.pos[0]) < 1e-4 and ch.pos[1] > 0.0
    assert abs(np.linalg.norm(ch.q[:]) - 1.0) < 1e-6
    for b in range(env_params.numBodies):
        assert np.all(np.array(s.body[b].lvel[:]) == 0) and np.all(np.array(s.body[b].avel[:]) == 0)
    assert s.currentGear == 1 or s.currentGear == 2   # neutral (index 1) before the autoshifter engages
    t = pc.DynState.from_buffer_copy(bytes(s))
    assert hostlib.pdb_teleport_to_spline(C.byref(env_params), flat_track, 0.5, C.byref(t)) == 0
    dz = t.body[0].pos[2] - s.body[0].pos[2]
    assert 1400.0 < dz < 1600.0
    # all bodies moved rigidly
    for b in range(env_params.numBodies):
        assert abs((t.body[b].pos[2] - s.body[b].pos[2]) - dz) < 1e-2
        assert abs(t.body[b].pos[1] - s.body[b].pos[1]) < 1e-3


def _fat_points(blob):
    nfat = struct.unpack_from('<8i', blob, 0)[4]
    off = struct.unpack_from('<7Q', blob, 48)[2]
    return np.frombuffer(blob, dtype='<f4', count=nfat * 15, offset=off).reshape(nfat, 15)


@needs_ref
@pytest.mark.parametrize('track', ['ebisu_touge', 'driftplayground', 'yamanashi_short', 'euphoria_hillside_park'])
def test_traced_sides_reproduce_the_shipped_spline_cache(hostlib, track):
    """Track::computeFatPoints + computeSideLocation (Sim/Track.cpp:366-467) recomputed from surfaces.bin + spline.bin
    against the spline.cache the reference saved from the same routine (Track.cpp:214-218) -- a reference-held golden for
    the track build step and, through the 2 x 1000 oblique rays per point, for the ray-vs-mesh query (ODE's collider in the
    reference).  The racing-line hits are identical; a side is the last accepted hit of a 1 cm fan, so a last-ulp
    difference at an acceptance threshold moves that side by a few steps -- seen at <= 2 % of the points of a track, <= 0.2 m."""
    blob = pc.build_track(hostlib, REF, track, recompute_fat_points=True)
    fat = _fat_points(blob)
    cache = np.fromfile(os.path.join(REF, 'content', 'tracks', track, 'spline.cache'), dtype='<f4').reshape(-1, 15)
    assert fat.shape == cache.shape
    assert np.array_equal(fat[:, 0:3], cache[:, 0:3])                       # best: vertical ray hits, bit for bit
    side_err = np.abs(fat[:, 3:9] - cache[:, 3:9]).max(axis=1)
    assert np.mean(side_err < 1e-4) > 0.98, np.mean(side_err < 1e-4)
    assert side_err.max() < 0.25, side_err.max()
    assert np.abs(fat[:, 12:15] - cache[:, 12:15]).max() < 1e-2
    # the default build (cache present) still takes the cached points verbatim, like Track::init
    assert np.array_equal(_fat_points(pc.build_track(hostlib, REF, track)), cache)


def test_traced_sides_on_a_synthetic_road(hostlib, tmp_path):
    """TRACE_SIDES=1 without any cache: the sides run out to the ribbon's edge (9 m, rays beyond it miss and are skipped),
    stay at the racing line where the road surface is not valid track, and stop at a bad sector's border."""
    import shutil, synthetic_tracks
    base = str(tmp_path)
    synthetic_tracks.make_base(base, tracks=('touge',))
    tdir = os.path.join(base, 'content', 'tracks', 'touge')

    def build(extra):
        with open(os.path.join(tdir, 'spline.ini'), 'w') as f:
            f.write('[SPLINE]\nCLOSED_LOOP=1\nTRACE_SIDES=1\n' + extra)
        return _fat_points(pc.build_track(hostlib, base, 'touge'))
    fat = build('TRACE_SIDE_MAX=12.0\n')
    w_l = np.linalg.norm(fat[:, 3:6] - fat[:, 0:3], axis=1); w_r = np.linalg.norm(fat[:, 6:9] - fat[:, 0:3], axis=1)
    kind = (np.arange(len(fat)) // 40) % len(synthetic_tracks.TOUGE_SURFACES)
    invalid = kind == 4                                                      # TOUGE_SURFACES[4]: valid=0
    inner = (np.arange(len(fat)) % 40 > 2) & (np.arange(len(fat)) % 40 < 38)  # away from the seams between surface kinds
    assert np.all(w_l[invalid & inner] == 0) and np.all(w_r[invalid & inner] == 0)
    hit = np.any(fat[:, 0:3] != 0, axis=1)                                  # a racing-line ray that slips through a mesh seam leaves the zeroed point (Track.cpp:389)
    ok = ~invalid & inner & hit
    assert ok.sum() > 400
    assert np.all((w_l[ok] > 8.7) & (w_l[ok] < 9.06)) and np.all((w_r[ok] > 8.7) & (w_r[ok] < 9.06)), (w_l[ok].min(), w_l[ok].max())
    assert np.allclose(fat[:, 9:12], 0.5 * (fat[:, 3:6] + fat[:, 6:9]), atol=1e-5)
    short = build('TRACE_SIDE_MAX=3.0\n')
    w = np.linalg.norm(short[:, 3:6] - short[:, 0:3], axis=1)
    assert np.all((w[ok] > 2.95) & (w[ok] < 3.03))                           # last step is (numSteps-1) * 1 cm
    bad = build('TRACE_SIDE_MAX=12.0\nTRACE_BAD_SECTORS=1|3\n')              # sector = surface index: points whose ray starts there stop at once
    wb = np.linalg.norm(bad[:, 3:6] - bad[:, 0:3], axis=1)
    sector = np.arange(len(fat)) // 40
    assert np.all(wb[inner & ((sector == 1) | (sector == 3))] == 0)
    assert np.all(wb[inner & (sector == 0)] > 8.7)


def test_pit_boxes_of_pits_ini(hostlib, base_dir):
    """Track::loadPits (reference Sim/Track.cpp:151-175): AC_PIT_0.. until the first missing section; the box's matrix = the rotation by ROT.x degrees about +y
    with POS in its fourth row.  The synthetic mountain road's pits.ini (five boxes) in the blob; pdb_teleport_to_pit puts the car into box `id` facing the
    box's heading and leaves it alone for an id outside the list; the shipped tracks' files where the reference content is present."""
    import synthetic_tracks, re
    d = os.path.join(base_dir, 'content', 'tracks', 'touge')
    synthetic_tracks.gen_touge(d)
    blob = pc.build_track(hostlib, base_dir, 'touge')
    secs = re.findall(r'\[AC_PIT_(\d+)\]\s+POS=([^\n]+)\s+ROT=([^\n]+)', open(os.path.join(d, 'pits.ini')).read())
    assert hostlib.pdb_track_num_pits(blob) == len(secs) == 5
    P = car_params('ks_toyota_ae86_drift')
    for k, (idx, pos, rot) in enumerate(secs):
        pos = [np.float32(x) for x in pos.split(',')]; ang = np.float32(np.float32(rot.split(',')[0]) * np.float32(0.01745329251994329576923690768489))
        m = (C.c_float * 16)(); assert hostlib.pdb_track_pit(blob, k, m) == 0
        m = np.array(m[:], np.float32).reshape(4, 4)
        assert tuple(m[3, :3]) == tuple(pos) and m[3, 3] == 1 and m[1, 1] == 1
        assert abs(m[2, 0] - np.sin(ang)) < 1e-6 and abs(m[2, 2] - np.cos(ang)) < 1e-6 and m[0, 0] == m[2, 2] and m[0, 2] == -m[2, 0]
        assert np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3), atol=1e-6)
        S = pc.DynState(); assert hostlib.pdb_initial_state(C.byref(P), blob, C.byref(S)) == 0
        assert hostlib.pdb_teleport_to_pit(C.byref(P), blob, k, C.byref(S)) == 0
        R = np.array(S.body[0].R[:], np.float32)
        assert abs(S.body[0].pos[0] - pos[0]) < 1e-5 and abs(S.body[0].pos[2] - pos[2]) < 1e-5
        # Car::forceRotation (Car.cpp:1274-1308) builds the body's matrix from the heading h = (M31, M32, M33): third row = h, first row = (h.z, 0, -h.x) / |.|
        Mb = R.reshape(3, 3)
        fwd = min((Mb[:, 2], Mb[2, :]), key=lambda v: np.abs(v - m[2, :3]).max())
        assert np.abs(fwd - m[2, :3]).max() < 1e-5
    S = pc.DynState(); assert hostlib.pdb_initial_state(C.byref(P), blob, C.byref(S)) == 0
    before = bytes(S)
    assert hostlib.pdb_teleport_to_pit(C.byref(P), blob, 5, C.byref(S)) == 0 and hostlib.pdb_teleport_to_pit(C.byref(P), blob, -1, C.byref(S)) == 0
    assert bytes(S) == before
    m = (C.c_float * 16)()
    assert hostlib.pdb_track_pit(blob, 5, m) != 0
    if os.path.isdir('/root/reference/content/tracks'):
        for trk in ('driftplayground', 'ebisu_touge', 'yamanashi_short', 'euphoria_hillside_park'):
            ids = set(int(x) for x in re.findall(r'\[AC_PIT_(\d+)\]', open('/root/reference/content/tracks/%s/pits.ini' % trk).read()))   # (ebisu_touge lists every box twice: the reader keeps the first)
            n = next(k for k in range(1000) if k not in ids)
            b = pc.build_track(hostlib, '/root/reference', trk)
            assert hostlib.pdb_track_num_pits(b) == n > 0, trk
