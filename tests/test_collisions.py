"""Body contacts: PhysicsEngineODE::collisionStep / onCollision + Car::onCollisionCallback --
detection, its consequences (collisionFlag, damage zones, drift validity, engine blow-up) and the response (contact joints in
the solve).  PARITY UNPINNED: ODE's colliders and its LCP solver are not in the reference tree, so contact generation and the
solution method follow this project's own definition; these tests check that definition's behaviour on
a walled strip (CPU oracle) and, under -m gpu, that the HIP path reproduces the oracle bit for bit."""
import ctypes as C, os
import numpy as np
import pytest
import pdb_ctypes as pc
import oracle_ctypes
from conftest import car_params

AE86 = 'ks_toyota_ae86_drift'


@pytest.fixture(scope='module')
def walled(hostlib):
    import synthetic_tracks, tempfile
    d = tempfile.mkdtemp(prefix='pdb_walled_')
    synthetic_tracks.make_base(d, tracks=('walled',))
    return pc.build_track(hostlib, d, 'walled')


def test_packed_cars_carry_their_colliders():
    """colliders.ini box + collider.bin hull of the shipped cars, in the chassis frame (graphics offset applied)"""
    for model, nv, nt in ((AE86, 50, 96), ('ks_mazda_rx7_tuned', 113, 178), ('ks_toyota_supra_mkiv_drift', 69, 102),
                          ('dthwsh_mazda_rx7_fc3s_sr20', 107, 170), ('gravygarage_street_ae86_readie', 50, 96)):
        c = car_params(model).collider
        assert (c.enabled, c.numBoxes, c.numVerts, c.numTris) == (1, 1, nv, nt)
        v = np.array([list(c.verts[i]) for i in range(nv)])
        t = np.array([list(c.tris[i]) for i in range(nt)])
        assert t.max() < nv
        assert np.allclose(v.min(0), list(c.boundsLo)[:3], atol=0.2) or c.boundsLo[1] < v[:, 1].min()    # the belly box hangs below the hull
        assert np.all(np.array(list(c.boundsLo)) <= np.minimum(v.min(0), np.array(list(c.boxCentre[0])) - np.array(list(c.boxHalf[0]))) + 1e-6)
        assert abs(v[:, 0].min() + v[:, 0].max()) < 1e-3 and 3.5 < v[:, 2].max() - v[:, 2].min() < 5.0     # symmetric, car-sized
        assert -0.45 < v[:, 1].min() < -0.15 and 0.6 < v[:, 1].max() < 1.0                               # sill .. roof around the CoG
    assert list(car_params(AE86).collider.boxHalf[0]) == pytest.approx([0.74, 0.05, 1.95])


def _drive(orc, hostlib, blob, P, ticks, shift_z=0.0, speed=0.0, frame0=0):
    """full throttle, steering centred; optionally start further down the strip, already rolling"""
    s0 = pc.DynState()
    assert hostlib.pdb_initial_state(C.byref(P), blob, C.byref(s0)) == 0
    for b in range(P.numBodies):
        s0.body[b].pos[2] += shift_z
        s0.body[b].lvel[2] = speed
    s0.simFrame = frame0
    h = orc.cpuref_create(C.byref(P), blob, len(blob), C.byref(s0))
    S = pc.DynState()
    rows = []
    for t in range(ticks):
        orc.cpuref_step_env(h, 0.0, 1.0)
        orc.cpuref_get_state(h, C.byref(S))
        rows.append((t, S.simFrame, S.collisionFlag, S.damageChanged, S.driftInvalid, S.body[0].pos[0], S.body[0].pos[2], S.speed * 3.6, S.lifeLeft) +
                    tuple(S.damageZoneLevel))
    orc.cpuref_destroy(h)
    return np.array(rows, dtype=np.float64)


def test_belly_box_scrapes_the_ridge(oracle, hostlib, walled):
    """the belly box (colliders.ini) meets TRACK surfaces only: over the 22 cm ridge the flag rises, on odd engine frames only
    (PhysicsEngineODE.cpp:230-241), and nothing is damaged (TRACK contacts carry no damage, Car.cpp:948-958)"""
    r = _drive(oracle, hostlib, walled, car_params(AE86), 1700)
    t, frame, flag, changed, invalid, x, z, kmh, life = r[:, :9].T
    assert np.all(frame == t + 1)
    hit = flag != 0
    assert hit.sum() >= 10
    assert np.all(frame[hit] % 2 == 0)                                   # the counter has already advanced: the collided frame was odd
    assert np.all(np.abs(z[hit] + 170.0) < 3.0)                          # only while the ridge is under the floor
    assert np.all(r[:, 9:] == 0) and np.all(changed == 0)


def test_nose_into_the_wall(oracle, hostlib, walled):
    """the hull (collider.bin) meets WALL surfaces: rolling at 54 km/h from 15 m out, the first contact comes when the nose
    (about 2 m ahead of the CoG) reaches the wall plane; front damage zone = closing speed in km/h (Car.cpp:964-1003), the
    drift is invalidated on the ticks the damage moves (ScoringSystem.cpp:360-368); the contact joints stop the car at the wall"""
    r = _drive(oracle, hostlib, walled, car_params(AE86), 900, shift_z=65.0, speed=15.0)
    t, frame, flag, changed, invalid, x, z, kmh, life = r[:, :9].T
    dmg = r[:, 9:]
    hit = flag != 0
    assert np.all(frame[hit] % 2 == 0)
    first = int(np.argmax(hit))
    assert abs(z[first] + 2.0 + 120.0) < 0.5, z[first]
    assert np.all(dmg[:first] == 0)
    assert abs(dmg[first, 0] - kmh[first]) < 0.05 * kmh[first] and dmg[first, 4] == dmg[first, 0] and np.all(dmg[first, 1:4] == 0)
    assert changed[first] == 1 and invalid[first] == 1
    assert np.all(np.diff(dmg, axis=0) >= 0)                             # zones only ever grow (tmax)
    assert np.all(life == life[0])                                       # < 150 km/h: the engine survives (Car.cpp:979-980)
    assert z.max() < -121.7                                              # the CoG never gets closer than the nose's length: held at the wall
    assert np.abs(x).max() < 1.5
    assert z[first + 10:first + 40].max() <= z[first + 9] + 0.02         # stopped within a few ticks of the first contact
    assert hit.sum() > 10                                                # pushed out, rolls back in under full throttle: again and again


def test_contact_response_comes_from_the_collision_pass(oracle, hostlib, walled):
    """with the collision pass switched off there are no contact joints either: the car drives through the wall"""
    P = car_params(AE86)
    on = _drive(oracle, hostlib, walled, P, 900, shift_z=65.0, speed=15.0)
    P.collider.enabled = 0
    off = _drive(oracle, hostlib, walled, P, 900, shift_z=65.0, speed=15.0)
    assert on[:, 2].sum() > 10 and off[:, 2].sum() == 0
    assert off[:, 6].max() > -112.0 and on[:, 6].max() < -121.7          # z: through and clear / held
    first = int(np.argmax(on[:, 2] != 0))
    assert np.array_equal(on[:first, 5:8], off[:first, 5:8])             # identical until the first contact
    assert np.all(off[:, 9:] == 0)


def test_blow_up_above_150_kmh(oracle, hostlib, walled):
    r = _drive(oracle, hostlib, walled, car_params(AE86), 60, shift_z=75.0, speed=47.2, frame0=1)
    assert r[-1, 8] == -100.0 and r[-1, 9] > 150.0                       # Engine::blowUp (Engine.cpp:406-409)


@pytest.mark.gpu
@pytest.mark.parametrize('model', [AE86, 'ks_toyota_supra_mkiv_drift', 'dthwsh_mazda_rx7_fc3s_sr20'])
def test_gpu_matches_oracle_on_the_walled_strip(built, model):
    """48 cars with their own constant steering fan out over ridge, side walls and cross wall: every state scalar incl. frame
    counter, flags and damage zones, every tick, bit for bit"""
    import parity_util
    seen = {'flag': 0, 'dmg': 0}

    def on_tick(t, i, sg, sc):
        seen['flag'] += int(sg.collisionFlag != 0)
        seen['dmg'] += int(sg.damageZoneLevel[4] > 0)
    worst = parity_util.run_parity(n_cars=48, ticks=2600, seed=99, track='walled', model=model, check_every=7, on_tick=on_tick)
    assert worst == 0.0, worst
    assert seen['flag'] > 100 and seen['dmg'] > 100, seen


@pytest.mark.gpu
@pytest.mark.parametrize('grid', ['1', '3'])
def test_contact_pass_results_do_not_depend_on_its_grid(built, grid, monkeypatch):
    """the contact pass's workgroups take the queued cars in turn: with one workgroup (or three) for dozens of cars, every group after
    the first runs in a workgroup that has already held other cars (stale LDS blocks, the pack wave's flag word, the staging block) --
    same bits as the oracle"""
    import parity_util
    monkeypatch.setenv('PDB_CONTACT_GRID', grid)   # read when the batch is created
    worst = parity_util.run_parity(n_cars=48, ticks=1400, seed=99, track='walled', model=AE86, check_every=7)
    assert worst == 0.0, worst


@pytest.mark.gpu
@pytest.mark.parametrize('form', ['one', 'pair', 'pair_grid1'])
@pytest.mark.parametrize('model', [AE86, 'dthwsh_mazda_rx7_fc3s_sr20'])
def test_contact_pass_forms_agree_with_the_oracle(built, form, model, monkeypatch):
    """the contact pass as ONE kernel (collision pass and back half of the tick by the car's wave) and as the PAIR of round 6 (collide: one car per workgroup, its
    narrow phase shared by four waves; resume: the first pass's workgroup shape, from the snapshot the collide kernel wrote back) -- the library picks per track
    (the fullest cell of the collision grid) and per launch (the hint); here each form is forced (PDB_CONTACT_SPLIT; a fixed grid makes the pair run on every tick, and
    with ONE workgroup every car after the first goes through LDS blocks that have held other cars): both classes of kernel (33 and 40 rows), bit for bit the oracle"""
    import parity_util
    monkeypatch.setenv('PDB_CONTACT_SPLIT', '0' if form == 'one' else '1')
    if form != 'one':
        monkeypatch.setenv('PDB_CONTACT_GRID', '1' if form == 'pair_grid1' else '32')
    seen = {'flag': 0, 'dmg': 0}

    def on_tick(t, i, sg, sc):
        seen['flag'] += int(sg.collisionFlag != 0)
        seen['dmg'] += int(sg.damageZoneLevel[4] > 0)
    worst = parity_util.run_parity(n_cars=48, ticks=1600, seed=99, track='walled', model=model, check_every=7, on_tick=on_tick)
    assert worst == 0.0, worst
    assert seen['flag'] > 20 and seen['dmg'] > 50, seen


@pytest.mark.gpu
@pytest.mark.parametrize('form', ['0', '1'])
def test_contact_pass_forms_agree_on_the_playground_scale_mesh(built, form, monkeypatch):
    """the same on the 118 320-triangle playground (grid cells with up to 196 entries: the long narrow phases the pair is for), cars spread over the lap"""
    monkeypatch.setenv('PDB_CONTACT_SPLIT', form)
    h, worst, seen = _scale_run('playground', AE86, 96, 1200, 11)
    assert worst == 0.0, worst
    assert seen['contacts'] > 200, seen


@pytest.mark.gpu
def test_gpu_matches_oracle_on_the_wall_lined_road(built):
    """the synthetic mountain road with guard rails (WALL surfaces along both edges, configs[4] shape): 32 cars with constant
    actions run wide into the rails within a few seconds -- hull contacts on a curved, banked, hilly mesh, bit for bit"""
    import parity_util, pdbatch
    blob = pdbatch.synthetic_track('touge', walls=True)
    seen = {'dmg': 0}

    def on_tick(t, i, sg, sc):
        seen['dmg'] += int(sg.damageZoneLevel[4] > 0)
    worst = parity_util.run_parity(n_cars=32, ticks=1800, seed=3, track=blob, check_every=9, on_tick=on_tick)
    assert worst == 0.0, worst
    assert seen['dmg'] > 10, seen


# ---- reference-scale collision meshes (BASELINE configs[4]; SURVEY 8d config 5: driftplayground = 510 surfaces / 112 411
# triangles / 490 WALL meshes; ks_nordschleife = 13 323 spline points).  The reference's own meshes do not travel to the GPU box;
# synthetic_tracks.gen_playground / gen_nordring build tracks of the same scale from closed-form geometry.
def _scale_run(track, model, n_cars, ticks, seed):
    import parity_util, pdbatch
    blob = pdbatch.reference_track(track) if track in pdbatch.REFERENCE_TRACKS else pdbatch.synthetic_track(track)
    h = pc.TrackHeader.from_buffer_copy(blob[:C.sizeof(pc.TrackHeader)])
    seen = {'flag': 0, 'dmg': 0, 'cars': set(), 'contacts': 0}

    def on_tick(t, i, sg, sc):
        seen['flag'] += int(sg.collisionFlag != 0)
        seen['contacts'] += int(sg.numContacts > 0)
        if sg.damageZoneLevel[4] > 0:
            seen['dmg'] += 1; seen['cars'].add(i)
    # cars spread over the whole lap by teleportCarToSpline, each with its own constant action: they leave the road within
    # seconds and meet barriers, tyre stacks, islands / the guard rails; no env mode, so nothing terminates on a hit
    worst = parity_util.run_parity(n_cars=n_cars, ticks=ticks, seed=seed, track=blob, model=model, check_every=8, on_tick=on_tick,
                                   spread=(0.0, 1.0), threads=8)
    return h, worst, seen


@pytest.mark.gpu
def test_gpu_matches_oracle_on_the_playground_scale_mesh(built):
    """24 cars x 1600 ticks on the synthetic paddock (510 surfaces, 118 k triangles, 490 separate WALL meshes, a 48 MB track blob --
    twelve times one XCD's L2): wheel rays, broad phase, hull-vs-obstacle contact points, the contact solve, damage -- every
    state scalar and every live contact joint bit for bit"""
    h, worst, seen = _scale_run('playground', AE86, 24, 1600, 7)
    assert h.numSurfaces >= 500 and h.numTris >= 100000
    assert worst == 0.0, worst
    assert len(seen['cars']) >= 12 and seen['contacts'] > 60, seen    # sampled every 8th tick


@pytest.mark.gpu
def test_gpu_matches_oracle_on_the_13k_point_walled_ribbon(built):
    """24 Supras x 1600 ticks spread over the 20.7 km open ribbon with guard rails (13 323 spline points, 1002 surfaces, 80 k
    triangles in a 6.5 km x 5.8 km box): the locator, the probes and the rails at Nordschleife scale"""
    h, worst, seen = _scale_run('nordring', 'ks_toyota_supra_mkiv_drift', 24, 1600, 7)
    assert h.numFat == 13323 and h.numTris >= 79000
    assert worst == 0.0, worst
    assert len(seen['cars']) >= 6 and seen['contacts'] > 40, seen     # sampled every 8th tick


# ---- the reference's OWN tracks, packed in the build container (tools/pack_tracks.py -> projectd-core_amd/data/tracks/)
@pytest.mark.gpu
def test_gpu_matches_oracle_on_driftplayground(built):
    """the env's default track (projectd_env.py:23; 510 surfaces, 112 411 triangles, 490 WALL meshes, Sim/Track.cpp:97-272):
    32 AE86s put down along the lap with their own constant actions for 1800 ticks -- within seconds they are in the barriers and
    tyre stacks; every state scalar and every live contact joint bit for bit"""
    h, worst, seen = _scale_run('driftplayground', AE86, 32, 1800, 11)
    assert h.numSurfaces == 510 and h.numTris == 112411
    assert worst == 0.0, worst
    assert len(seen['cars']) >= 8 and seen['contacts'] > 40, seen    # sampled every 8th tick


@pytest.mark.gpu
def test_gpu_matches_oracle_on_the_nordschleife_ribbon(built):
    """ks_nordschleife's shipped spline (13 323 points) with the road and the guard rails generated around it: 24 Supras x 1600 ticks"""
    h, worst, seen = _scale_run('ks_nordschleife_walls', 'ks_toyota_supra_mkiv_drift', 24, 1600, 5)
    assert h.numFat == 13323
    assert worst == 0.0, worst
    assert len(seen['cars']) >= 4 and seen['contacts'] > 20, seen


# ---- the contact path at BASELINE's sizes: configs[4]'s per-GPU shard is 8192 cars; the oracle cannot follow that many, replication can
@pytest.mark.gpu
@pytest.mark.parametrize('track,model,parts', [('driftplayground', AE86, 3), ('ks_nordschleife_walls', 'ks_toyota_supra_mkiv_drift', None)])
def test_full_size_contact_batches_by_replication(built, track, model, parts):
    """8192 cars = 32 different (start point, constant action) pairs x 256 replicas on the reference's own driftplayground mesh (three free-running partitions,
    as bench.py runs it) and on the walled Nordschleife ribbon, 1600 ticks without an env loop: the cars leave the road within seconds and STAY in the barriers,
    tyre stacks and rails, so thousands of cars sit in the contact pass at once -- its queue, the hand-over snapshots, the staging blocks, the 32-candidate cap,
    the adaptive grid.  Every replica byte-identical to its representative (record + live contact joints) wherever it sits; the representatives equal the oracle,
    contact joints included, every 50 ticks."""
    import parity_util
    r = parity_util.run_replicated(8192, 32, 1600, track, model=model, seed=11, check_every=50, partitions=parts)
    print('%s: worst %.3e, up to %d of 8192 cars with live contact joints at a check, %d representative checks with live joints' % (track, r['worst'], r['max_in_contact'], r['contact_checks']))
    assert r['worst'] == 0.0, r
    assert r['max_in_contact'] >= 512 and r['contact_checks'] >= 10, r


@pytest.mark.gpu
def test_host_fed_policy_at_full_size_by_replication(built):
    """BASELINE configs[4] as worded on one GPU's shard: 8192 cars on the reference's driftplayground (walls), "SAC-policy actions fed from host" every tick -- observations
    down, a float32 24-256-256-2 MLP (the bench's fixed random actor, seed 4567) on the host, actions up -- through the library's pipelined form over three free-running
    partitions (pdb_step_host_partition / pdb_wait_host_partition).  32 representatives tiled 256 times: every replica's output row equal to its representative's every
    tick, every replica's record and live contact joints byte-identical at every check, the representatives equal to the oracle fed the same law."""
    import parity_util
    rs = np.random.RandomState(4567)
    w1 = (rs.randn(24, 256) / 24 ** 0.5).astype(np.float32); w2 = (rs.randn(256, 256) / 16.0).astype(np.float32); w3 = (rs.randn(256, 2) / 16.0).astype(np.float32)
    b3 = np.array([0.0, 0.5], np.float32)
    import projectd_env
    scale = (1.0 / projectd_env.obs_bounds(projectd_env.EnvConfig())[1]).astype(np.float32)

    def law(obs, t, ids):   # float32 throughout, the same shapes for both sides (the law sees the representatives' rows only)
        h = np.tanh((obs * scale) @ w1, dtype=np.float32)
        h = np.tanh(h @ w2, dtype=np.float32)
        a = np.tanh(h @ w3 + b3, dtype=np.float32)
        a[:, 0] += np.float32(0.7) * ((ids % 3) - 1).astype(np.float32)   # an untrained actor that also pulls two thirds of the cars to one side or the other: they meet the walls
        a[:, 1] += np.float32(0.5)
        return np.clip(a, -1.0, 1.0).astype(np.float32)
    r = parity_util.run_replicated(8192, 32, 1200, 'driftplayground', law=law, partitions=3, check_every=100, host_pipeline=True)
    assert r['worst'] == 0.0, r
    assert r['max_in_contact'] >= 256 and r['contact_checks'] > 0, r   # cars really were leaning on the walls while the policy drove them
