"""The track blob's acceleration structures against brute force, on the CPU (the kernel's results are compared with an oracle that
scans everything; this pins the structures themselves, in particular the conservative binning of the wheel rays' grid):
* every triangle a vertical ray hits (the kernel's float32 Moeller-Trumbore, restated in numpy) is listed in the ray's cell of the
  ray grid, and in the ray's cell of the collision grid;
* ray records are copies of the triangles, in ascending triangle order per cell;
* every fat point within probe range of a query point is listed in the fat-point cells under the query box;
* the per-entry position + id records and the per-point side-segment records equal the arrays they are derived from."""
import ctypes as C, os, tempfile
import numpy as np
import pytest
import pdb_ctypes as pc


def arrays(blob):
    h = pc.TrackHeader.from_buffer_copy(blob[:C.sizeof(pc.TrackHeader)])
    assert h.version == 6 and h.totalBytes == len(blob)
    buf = np.frombuffer(blob, dtype=np.uint8)
    f = lambda off, n: buf[off:off + 4 * n].view(np.float32)
    i = lambda off, n: buf[off:off + 4 * n].view(np.int32)
    tris = f(h.offTris, 9 * h.numTris).reshape(-1, 9)
    fat = f(h.offFat, 15 * h.numFat).reshape(-1, 15)
    gs = i(h.offGridStart, h.gridNx * h.gridNz + 1); gt = i(h.offGridTris, int(gs[-1]))
    rs = i(h.offRayStart, h.rayNx * h.rayNz + 1); rr = buf[h.offRayRecs:h.offRayRecs + 48 * int(rs[-1])].view(np.float32).reshape(-1, 12)
    fgs = i(h.offFatGridStart, h.fatGridNx * h.fatGridNz + 1); fgi = i(h.offFatGridIds, h.numFat)
    fgr = f(h.offFatGridRec, 4 * h.numFat).reshape(-1, 4); seg = f(h.offFatSeg, 8 * h.numFat).reshape(-1, 8)
    return h, tris, fat, gs, gt, rs, rr, fgs, fgi, fgr, seg


def ray_hits(tris, ox, oy, oz, max_dist=3.0):
    """rayTriDev for d = (0, -1, 0), float32 like the kernel: ids of the triangles hit"""
    f32 = np.float32
    v0, v1, v2 = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
    e1 = v1 - v0; e2 = v2 - v0
    d = np.array([0, -1, 0], f32)
    p = np.cross(np.broadcast_to(d, e2.shape), e2).astype(f32)
    det = (e1 * p).sum(1, dtype=f32)
    tv = np.array([ox, oy, oz], f32) - v0
    u = (tv * p).sum(1, dtype=f32)
    q = np.cross(tv, e1).astype(f32)
    v = (q * d).sum(1, dtype=f32)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = (e2 * q).sum(1, dtype=f32) * (f32(1.0) / det)
    ok = (det >= f32(1e-6)) & (u >= 0) & (u <= det) & (v >= 0) & (u + v <= det) & (t >= 0) & (t < f32(max_dist))
    return np.nonzero(ok)[0]


REF = '/root/reference'


@pytest.mark.parametrize('kind,gen', [('touge', {'step': 0.9}), ('hillclimb', {}), ('walled', {}), ('playground', {}), ('nordring', {}),
                                      ('ref:driftplayground', {})])
def test_grids_list_everything_brute_force_finds(hostlib, kind, gen):
    import synthetic_tracks
    if kind.startswith('ref:'):     # the reference's own 112 411-triangle mesh (build container only)
        if not os.path.isdir(os.path.join(REF, 'content', 'tracks', kind[4:])):
            pytest.skip('reference content not present')
        blob = pc.build_track(hostlib, REF, kind[4:])
    else:
        d = tempfile.mkdtemp(prefix='pdb_grid_')
        synthetic_tracks.make_base(d, tracks=())
        synthetic_tracks.GENERATORS[kind](os.path.join(d, 'content', 'tracks', 't'), **gen)
        blob = pc.build_track(hostlib, d, 't')
    h, tris, fat, gs, gt, rs, rr, fgs, fgi, fgr, seg = arrays(blob)
    # records = copies, ascending per cell
    ids = rr[:, 9].view(np.int32)
    assert np.array_equal(rr[:, :9], tris[ids])
    for c in np.random.default_rng(1).integers(0, h.rayNx * h.rayNz, 400):
        run = ids[rs[c]:rs[c + 1]]
        assert np.all(np.diff(run) > 0)
    # derived probe records
    assert np.array_equal(fgr[:, 3].view(np.int32), fgi) and np.array_equal(fgr[:, :3], fat[fgi, 0:3])
    nxt = np.roll(np.arange(h.numFat), -1)
    assert np.array_equal(seg, np.stack([fat[:, 3], fat[:, 5], fat[nxt, 3], fat[nxt, 5], fat[:, 6], fat[:, 8], fat[nxt, 6], fat[nxt, 8]], 1))
    # rays: on and around the road (points of the fat spline, jittered out to the edges and beyond), and on cell boundaries
    rng = np.random.default_rng(7)
    f32 = np.float32
    pts = []
    for k in rng.integers(0, h.numFat, 150):
        c = fat[k, 0:3]
        pts.append((f32(c[0] + rng.uniform(-12, 12)), f32(c[1] + 2.0), f32(c[2] + rng.uniform(-12, 12))))
    for k in rng.integers(0, h.numFat, 60):   # exactly on a ray-grid line
        c = fat[k, 0:3]
        gx = f32(h.rayMinX) + f32(np.floor((c[0] - h.rayMinX) / h.rayCell)) * f32(h.rayCell)
        pts.append((f32(gx), f32(c[1] + 2.0), f32(c[2])))
    nhit = 0
    for (ox, oy, oz) in pts:
        hit = ray_hits(tris, ox, oy, oz)
        nhit += len(hit)
        ix = int(np.floor((f32(ox) - f32(h.rayMinX)) / f32(h.rayCell))); iz = int(np.floor((f32(oz) - f32(h.rayMinZ)) / f32(h.rayCell)))
        if len(hit):
            assert 0 <= ix < h.rayNx and 0 <= iz < h.rayNz
            c = iz * h.rayNx + ix
            assert set(hit.tolist()) <= set(ids[rs[c]:rs[c + 1]].tolist()), (ox, oz)
            jx = int(np.floor((f32(ox) - f32(h.gridMinX)) / f32(h.gridCell))); jz = int(np.floor((f32(oz) - f32(h.gridMinZ)) / f32(h.gridCell)))
            c2 = jz * h.gridNx + jx
            assert set(hit.tolist()) <= set(gt[gs[c2]:gs[c2 + 1]].tolist()), (ox, oz)
    assert nhit > 50
    # broad phase of the collision pass: every triangle whose box meets a chassis-sized AABB is listed in the cells under that AABB
    tmin = np.minimum(np.minimum(tris[:, 0:3], tris[:, 3:6]), tris[:, 6:9]); tmax = np.maximum(np.maximum(tris[:, 0:3], tris[:, 3:6]), tris[:, 6:9])
    met = 0
    for k in rng.integers(0, h.numFat, 60):
        c = fat[k, 0:3] + rng.uniform(-15, 15, 3).astype(f32) * np.array([1, 0.05, 1], f32)
        lo = (c - np.array([2.6, 0.9, 2.6], f32)).astype(f32); hi = (c + np.array([2.6, 0.9, 2.6], f32)).astype(f32)
        want = np.nonzero(np.all(tmin <= hi, 1) & np.all(tmax >= lo, 1))[0]
        met += len(want)
        cell = lambda v, mn: np.floor((f32(v) - f32(mn)) / f32(h.gridCell))
        x0 = int(min(max(cell(lo[0], h.gridMinX), 0), h.gridNx)); x1 = int(min(max(cell(hi[0], h.gridMinX), -1), h.gridNx - 1))
        z0 = int(min(max(cell(lo[2], h.gridMinZ), 0), h.gridNz)); z1 = int(min(max(cell(hi[2], h.gridMinZ), -1), h.gridNz - 1))
        listed = set()
        for z in range(z0, z1 + 1):
            if x0 <= x1:
                listed |= set(gt[gs[z * h.gridNx + x0]:gs[z * h.gridNx + x1 + 1]].tolist())
        assert set(want.tolist()) <= listed, (lo, hi)
    assert met > 100
    # fat points within probe range (50 m) of a query point are all in the cells under the query box
    for k in rng.integers(0, h.numFat, 40):
        q = fat[k, 0:3] + rng.uniform(-20, 20, 3).astype(f32)
        near = np.nonzero(((fat[:, 0:3] - q) ** 2).sum(1) < 50.0 ** 2)[0]
        reach = 50.01
        x0 = int(np.clip(np.floor((q[0] - reach - h.fatGridMinX) / h.fatGridCell), 0, h.fatGridNx)); x1 = int(np.clip(np.floor((q[0] + reach - h.fatGridMinX) / h.fatGridCell), -1, h.fatGridNx - 1))
        z0 = int(np.clip(np.floor((q[2] - reach - h.fatGridMinZ) / h.fatGridCell), 0, h.fatGridNz)); z1 = int(np.clip(np.floor((q[2] + reach - h.fatGridMinZ) / h.fatGridCell), -1, h.fatGridNz - 1))
        listed = set()
        for z in range(z0, z1 + 1):
            if x0 <= x1:
                listed |= set(fgi[fgs[z * h.fatGridNx + x0]:fgs[z * h.fatGridNx + x1 + 1]].tolist())
        assert set(near.tolist()) <= listed
