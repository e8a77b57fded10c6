#!/usr/bin/env python3
"""Synthetic benchmark / test tracks written in the reference's on-disk formats:
surfaces.bin (58-byte packed BlobSurface header + float[3] verts + uint16 indices,
Sim/Surface.h:25-45, Sim/Track.cpp:97-150), spline.bin (SlimTrackPoint = float[3] best +
float[2] sides, Sim/Track.h:12-17), spline.ini (Sim/Track.cpp:158-186).

Tracks:
  flat    : one 2-triangle TRACK surface spanning +-2 km at y=0, straight spline along +z every
            10 m with 6 m sides.
"""
import os, struct, sys, math

MAGIC = 0xAABBCCDD

def write_surface(f, verts, indices, sector=0, category=1, grip=1.0, valid=1, damping=0.0, sin_height=0.0, sin_length=0.0,
                  granularity=0.0, dirt=0.0):
    # BlobSurface (Sim/Surface.h:25-45): gripMod, damping, sinHeight, sinLength, granularity, dirtAdditiveK, vibrationGain,
    # vibrationLength, wavPitchSpeed, isValidTrack, isPitlane
    hdr = struct.pack('<5I9f2B', MAGIC, len(verts), len(indices), sector, category,
                      grip, damping, sin_height, sin_length, granularity, dirt, 0.0, 0.0, 0.0, valid, 0)
    assert len(hdr) == 58
    f.write(hdr)
    for v in verts:
        f.write(struct.pack('<3f', *v))
    f.write(struct.pack('<%dH' % len(indices), *indices))

def gen_flat(out, half=2000.0, z0=-1500.0, z1=1500.0, step=10.0, side=6.0):
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'surfaces.bin'), 'wb') as f:
        # (v1-v0)x(v2-v0) = +y for both triangles (front face seen from above)
        verts = [(-half, 0.0, -half), (-half, 0.0, half), (half, 0.0, half), (half, 0.0, -half)]
        write_surface(f, verts, [0, 1, 2, 0, 2, 3])
    n = int(round((z1 - z0) / step)) + 1
    with open(os.path.join(out, 'spline.bin'), 'wb') as f:
        for i in range(n):
            f.write(struct.pack('<5f', 0.0, 0.0, z0 + step * i, side, side))
    with open(os.path.join(out, 'spline.ini'), 'w') as f:
        f.write('[SPLINE]\nCLOSED_LOOP=0\nTRACE_SIDES=0\n')
    # remove stale fat-point cache so the loader recomputes it
    c = os.path.join(out, 'spline.cache')
    if os.path.exists(c):
        os.remove(c)

def gen_walled(out, z0=-200.0, z1=200.0, step=10.0, side=6.0, half_width=8.0, wall_z=-120.0, bump_z=-170.0, bump_h=0.22):
    """Walled test strip (BASELINE configs[4] shape in miniature: a collision mesh next to the road): the flat plane, WALL
    surfaces (category 2) on both sides at x = +-half_width and across the road at wall_z, and a 0.22 m ridge of TRACK
    surface across the road at bump_z that the wheels climb and the belly box scrapes.  The car starts at z0 heading +z."""
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'surfaces.bin'), 'wb') as f:
        half = 1000.0
        write_surface(f, [(-half, 0.0, -half), (-half, 0.0, half), (half, 0.0, half), (half, 0.0, -half)], [0, 1, 2, 0, 2, 3])
        # ridge: two slopes meeting at height bump_h, 0.6 m run each (normal.y = 0.94), front faces up
        x = half_width + 2.0
        rv = [(-x, 0.0, bump_z - 0.6), (x, 0.0, bump_z - 0.6), (-x, bump_h, bump_z), (x, bump_h, bump_z), (-x, 0.0, bump_z + 0.6), (x, 0.0, bump_z + 0.6)]
        write_surface(f, rv, [0, 2, 1, 1, 2, 3, 2, 4, 3, 3, 4, 5], sector=1, grip=0.95)
        for sgn in (-1.0, 1.0):     # side walls in 10 m panels
            verts, idx = [], []
            z = z0 - 10.0
            while z < wall_z + 10.0:
                b = len(verts)
                verts += [(sgn * half_width, -0.5, z), (sgn * half_width, 2.0, z), (sgn * half_width, 2.0, z + 10.0), (sgn * half_width, -0.5, z + 10.0)]
                idx += [b, b + 1, b + 2, b, b + 2, b + 3]
                z += 10.0
            write_surface(f, verts, idx, sector=2, category=2, valid=0)
        verts, idx = [], []
        for k in range(4):          # the wall across the road, four panels
            xa = -half_width + k * (half_width / 2.0); xb = xa + half_width / 2.0
            b = len(verts)
            verts += [(xa, -0.5, wall_z), (xa, 2.0, wall_z), (xb, 2.0, wall_z), (xb, -0.5, wall_z)]
            idx += [b, b + 1, b + 2, b, b + 2, b + 3]
        write_surface(f, verts, idx, sector=3, category=2, valid=0)
    n = int(round((z1 - z0) / step)) + 1
    with open(os.path.join(out, 'spline.bin'), 'wb') as f:
        for i in range(n):
            f.write(struct.pack('<5f', 0.0, 0.0, z0 + step * i, side, side))
    with open(os.path.join(out, 'spline.ini'), 'w') as f:
        f.write('[SPLINE]\nCLOSED_LOOP=0\nTRACE_SIDES=0\n')
    c = os.path.join(out, 'spline.cache')
    if os.path.exists(c):
        os.remove(c)

def touge_centreline(step=5.0, radius=600.0):
    """Closed mountain-road centreline: a wobbly ring (3- and 5-lobed) with +-40 m of elevation, resampled at `step` metres.
    Pure float64 math with fixed constants => the same bytes on every machine."""
    m = 20000
    pts = []
    for k in range(m):
        th = 2.0 * math.pi * k / m
        r = radius * (1.0 + 0.25 * math.sin(3.0 * th) + 0.10 * math.sin(5.0 * th + 1.0))
        pts.append((r * math.cos(th), 30.0 * math.sin(2.0 * th) + 10.0 * math.sin(7.0 * th + 0.5), r * math.sin(th)))
    cum = [0.0]
    for k in range(m):
        a, b = pts[k], pts[(k + 1) % m]
        cum.append(cum[-1] + math.sqrt(sum((b[i] - a[i]) ** 2 for i in range(3))))
    total = cum[-1]
    n = int(round(total / step))
    out = []
    j = 0
    for i in range(n):
        s = total * i / n
        while cum[j + 1] < s:
            j += 1
        t = (s - cum[j]) / (cum[j + 1] - cum[j])
        a, b = pts[j], pts[(j + 1) % m]
        out.append(tuple(a[c] + t * (b[c] - a[c]) for c in range(3)))
    return out


# surface kinds cycled along the road (values in the range of the reference's shipped surfaces.ini files): road, smooth,
# rumble strip (sine-wave height), road, grass-like (low grip, damping, dirt), granular
TOUGE_SURFACES = [dict(grip=0.97), dict(grip=0.98), dict(grip=0.95, sin_height=0.03, sin_length=0.5), dict(grip=0.97),
                  dict(grip=0.8, damping=0.01, dirt=1.0, valid=0), dict(grip=0.9, granularity=1.0, dirt=0.1)]


def gen_touge(out, step=5.0, side=5.0, margin=4.0, bank_gain=6.0, bank_max=0.08, points_per_surface=40, walls=False):
    """Synthetic "Akina-like" closed mountain road (BASELINE configs[2] shape): curvy ring with hills and curvature-
    proportional banking; ribbon mesh (two triangles per spline interval) cut into surfaces of points_per_surface intervals
    that cycle through six kinds of surface properties, spline every `step` metres with symmetric sides, CLOSED_LOOP=1.
    walls=True lines both edges of the ribbon with a 1.3 m WALL surface (category 2): BASELINE configs[4] shape, a wall-lined road."""
    os.makedirs(out, exist_ok=True)
    c = touge_centreline(step)
    n = len(c)
    lat, bank = [], []
    for i in range(n):
        p0, p1, p2 = c[i - 1], c[i], c[(i + 1) % n]
        fx, fz = p2[0] - p0[0], p2[2] - p0[2]
        fl = math.hypot(fx, fz)
        fx, fz = fx / fl, fz / fl
        lat.append((fz, -fx))                      # unit lateral vector in the xz plane (to the right of travel for +z forward... sign is irrelevant: symmetric)
        ax, az = p1[0] - p0[0], p1[2] - p0[2]
        bx, bz = p2[0] - p1[0], p2[2] - p1[2]
        turn = (ax * bz - az * bx) / (math.hypot(ax, az) * math.hypot(bx, bz))   # sin of the heading change per interval
        kappa = turn / step
        bank.append(max(-bank_max, min(bank_max, bank_gain * kappa)))
    half = side + margin

    def edge(i, sgn):
        p = c[i]; l = lat[i]
        return (p[0] + sgn * half * l[0], p[1] - sgn * half * math.sin(bank[i]), p[2] + sgn * half * l[1])
    with open(os.path.join(out, 'surfaces.bin'), 'wb') as f:
        i0 = 0
        while i0 < n:
            cnt = min(points_per_surface, n - i0)
            verts, idx = [], []
            for k in range(cnt + 1):
                i = (i0 + k) % n
                verts.append(edge(i, -1.0)); verts.append(edge(i, +1.0))
            for k in range(cnt):
                a, b, cc, d = 2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 3          # a,b at interval start (left,right); cc,d at its end
                for tri in ((a, cc, d), (a, d, b)):
                    v0, v1, v2 = verts[tri[0]], verts[tri[1]], verts[tri[2]]
                    ny = (v1[2] - v0[2]) * (v2[0] - v0[0]) - (v1[0] - v0[0]) * (v2[2] - v0[2])
                    idx.extend(tri if ny > 0 else (tri[0], tri[2], tri[1]))      # front face up
            write_surface(f, [tuple(float(x) for x in v) for v in verts], idx, sector=i0 // points_per_surface,
                          **TOUGE_SURFACES[(i0 // points_per_surface) % len(TOUGE_SURFACES)])
            if walls:
                for sgn in (-1.0, 1.0):
                    wv, wi = [], []
                    for k in range(cnt + 1):
                        e = edge((i0 + k) % n, sgn)
                        wv.append((float(e[0]), float(e[1]) - 0.3, float(e[2]))); wv.append((float(e[0]), float(e[1]) + 1.0, float(e[2])))
                    for k in range(cnt):
                        a, b, cc, d = 2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 3
                        wi.extend((a, b, d, a, d, cc))
                    write_surface(f, wv, wi, sector=1000 + i0 // points_per_surface, category=2, valid=0)
            i0 += cnt
    with open(os.path.join(out, 'spline.bin'), 'wb') as f:
        for p in c:
            f.write(struct.pack('<5f', p[0], p[1], p[2], side, side))
    with open(os.path.join(out, 'spline.ini'), 'w') as f:
        f.write('[SPLINE]\nCLOSED_LOOP=1\nTRACE_SIDES=0\n')
    cch = os.path.join(out, 'spline.cache')
    if os.path.exists(cch):
        os.remove(cch)
    return n


def read_spline_bin(path):
    """spline.bin: SlimTrackPoint = float[3] best + float[2] sides, no header (Sim/Track.cpp:97-150)"""
    raw = open(path, 'rb').read()
    n = len(raw) // 20
    return [struct.unpack_from('<5f', raw, 20 * i) for i in range(n)]


def gen_ribbon(out, points, closed=False, margin=4.0, points_per_surface=40, lead=3):
    """Road ribbon extruded around a given spline (list of (x, y, z, side_left, side_right), e.g. read_spline_bin of a track that
    ships its spline but not its mesh): two triangles per spline interval, flat cross-section at the point's height, half widths
    = the point's sides + margin, surfaces of points_per_surface intervals cycling through TOUGE_SURFACES.  An open spline gets
    `lead` extrapolated intervals before its first and after its last point (the start pose must have road under the rear
    wheels).  Writes surfaces.bin only: the spline files stay the caller's."""
    os.makedirs(out, exist_ok=True)
    pts = [tuple(float(v) for v in p) for p in points]
    if not closed:
        a, b = pts[0], pts[1]
        head = [tuple(a[c] + (a[c] - b[c]) * k for c in range(3)) + (a[3], a[4]) for k in range(lead, 0, -1)]
        a, b = pts[-1], pts[-2]
        tail = [tuple(a[c] + (a[c] - b[c]) * k for c in range(3)) + (a[3], a[4]) for k in range(1, lead + 1)]
        pts = head + pts + tail
    n = len(pts)
    lat = []
    for i in range(n):
        p0 = pts[i - 1] if (closed or i > 0) else pts[i]
        p2 = pts[(i + 1) % n] if (closed or i + 1 < n) else pts[i]
        fx, fz = p2[0] - p0[0], p2[2] - p0[2]
        fl = math.hypot(fx, fz)
        lat.append((fz / fl, -fx / fl) if fl > 0.0 else (lat[-1] if lat else (1.0, 0.0)))

    def edge(i, sgn):
        p = pts[i]; l = lat[i]
        w = (p[3] if sgn < 0 else p[4]) + margin
        return (p[0] + sgn * w * l[0], p[1], p[2] + sgn * w * l[1])
    nseg = n if closed else n - 1
    with open(os.path.join(out, 'surfaces.bin'), 'wb') as f:
        i0 = 0
        while i0 < nseg:
            cnt = min(points_per_surface, nseg - i0)
            verts, idx = [], []
            for k in range(cnt + 1):
                i = (i0 + k) % n
                verts.append(edge(i, -1.0)); verts.append(edge(i, +1.0))
            for k in range(cnt):
                a, b, cc, d = 2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 3
                for tri in ((a, cc, d), (a, d, b)):
                    v0, v1, v2 = verts[tri[0]], verts[tri[1]], verts[tri[2]]
                    ny = (v1[2] - v0[2]) * (v2[0] - v0[0]) - (v1[0] - v0[0]) * (v2[2] - v0[2])
                    idx.extend(tri if ny > 0 else (tri[0], tri[2], tri[1]))      # front face up
            write_surface(f, verts, idx, sector=i0 // points_per_surface, **TOUGE_SURFACES[(i0 // points_per_surface) % len(TOUGE_SURFACES)])
            i0 += cnt
    return n


def gen_hillclimb(out, step=0.9, length=4300.0, side=5.0, **kw):
    """Synthetic stand-in for the reference's ek_akina (5109 points, ~0.9 m, open): the first `length` metres of the
    mountain-road centreline as an OPEN spline with uneven point spacing (+-40 %), the best point wandering +-2 m across the road
    like a racing line and the two sides asymmetric accordingly; road = gen_ribbon around it."""
    os.makedirs(out, exist_ok=True)
    fine = touge_centreline(0.1)
    pts = []
    s, k = 0.0, 0
    length = min(length, 0.1 * (len(fine) - 2))
    while s < length:
        i = int(s / 0.1)
        p0, p1 = fine[i - 1], fine[(i + 1) % len(fine)]
        fx, fz = p1[0] - p0[0], p1[2] - p0[2]
        fl = math.hypot(fx, fz)
        lx, lz = fz / fl, -fx / fl
        r = 2.0 * math.sin(s / 40.0)
        c = fine[i]
        pts.append((c[0] + r * lx, c[1], c[2] + r * lz, side + r, side - r))
        s += step * (1.0 + 0.4 * math.sin(0.7 * k))
        k += 1
    with open(os.path.join(out, 'spline.bin'), 'wb') as f:
        for p in pts:
            f.write(struct.pack('<5f', *p))
    with open(os.path.join(out, 'spline.ini'), 'w') as f:
        f.write('[SPLINE]\nCLOSED_LOOP=0\nTRACE_SIDES=0\n')
    cch = os.path.join(out, 'spline.cache')
    if os.path.exists(cch):
        os.remove(cch)
    gen_ribbon(out, read_spline_bin(os.path.join(out, 'spline.bin')), closed=False, **kw)
    return len(pts)


def ribbon_track_from(src_track_dir, out, **kw):
    """A loadable track directory for a track whose mesh is not at hand: its own spline.bin / spline.ini / pits.ini copied, the
    road generated around the spline (gen_ribbon).  The reference's ek_akina and ks_nordschleife come this way."""
    import shutil
    os.makedirs(out, exist_ok=True)
    for fn in ('spline.bin', 'spline.ini', 'pits.ini'):
        sp = os.path.join(src_track_dir, fn)
        if os.path.exists(sp):
            shutil.copyfile(sp, os.path.join(out, fn))
    closed = False
    ini = os.path.join(out, 'spline.ini')
    if os.path.exists(ini):
        for ln in open(ini):
            if ln.strip().upper().startswith('CLOSED_LOOP'):
                closed = ln.split('=')[1].strip().split()[0] not in ('0', '')
    cch = os.path.join(out, 'spline.cache')
    if os.path.exists(cch):
        os.remove(cch)
    return gen_ribbon(out, read_spline_bin(os.path.join(out, 'spline.bin')), closed=closed, **kw)


SIM_INI = '''[SIM]
STEP_HZ=333
MAX_CARS=2

[ENVIRONMENT]
ROAD_TEMP=20.0
AMBIENT_TEMP=20.0
TRACK_GRIP=0.98

[VERTEX_HASH]
CELL_SIZE=50.0
TABLE_SIZE=4096

[CAR_LOOK_AHEAD]
COUNT=5
STEP=10.0

[CAR_PROBE_1]
YAW=0.0
LENGTH=50.0

[CAR_PROBE_2]
YAW=25.0
LENGTH=50.0

[CAR_PROBE_3]
YAW=-25.0
LENGTH=50.0

[CAR_PROBE_4]
YAW=90.0
LENGTH=10.0

[CAR_PROBE_5]
YAW=-90.0
LENGTH=10.0

[CAR_PROBE_6]
YAW=-145.0
LENGTH=10.0

[CAR_PROBE_7]
YAW=145.0
LENGTH=10.0
'''

def install_packed_car(base, model='ks_toyota_ae86_drift', block='ks_toyota_ae86_drift.env'):
    """Put one of the package's packed car blocks where the loader looks for cars that ship without their INI data:
    <base>/content/cars/<model>/<model>.pdcar"""
    import shutil
    dst = os.path.join(base, 'content', 'cars', model)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', block + '.pdcar'), os.path.join(dst, model + '.pdcar'))


def make_base(base, tracks=('flat',)):
    """Create <base>/cfg/sim.ini (same keys/values as the reference's shipped cfg/sim.ini that the
    hot path reads: Sim/Simulator.cpp:62-75, Sim/Track.cpp:38-42,212-216, Car/Car.cpp:285-314) and the
    requested synthetic tracks under <base>/content/tracks/."""
    os.makedirs(os.path.join(base, 'cfg'), exist_ok=True)
    with open(os.path.join(base, 'cfg', 'sim.ini'), 'w') as f:
        f.write(SIM_INI)
    for t in tracks:
        {'flat': gen_flat, 'touge': gen_touge, 'walled': gen_walled, 'hillclimb': gen_hillclimb}[t](os.path.join(base, 'content', 'tracks', t))
    return base

if __name__ == '__main__':
    kind, out = sys.argv[1], sys.argv[2]
    {'flat': gen_flat, 'touge': gen_touge, 'walled': gen_walled, 'hillclimb': gen_hillclimb}[kind](out)
