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
    # pit boxes in the shipped tracks' format (pits.ini: AC_PIT_n sections, POS, ROT in degrees with the heading in ROT.x; Sim/Track.cpp:151-175): on the road, to either
    # side of its centre, one 40 m off the ribbon with nothing under it; headings of both signs and beyond a full turn
    with open(os.path.join(out, 'pits.ini'), 'w') as f:
        for k, (frac, off, up, rot) in enumerate(((1.0 / 7.0, 2.0, 0.8, 122.2), (1.0 / 3.0, -3.0, 0.4, -47.5), (0.5, 0.0, 1.5, 301.25), (2.0 / 3.0, 40.0, 0.5, 15.0),
                                                  (0.9, -1.5, 0.2, 400.0))):
            i = int(frac * n)
            f.write('[AC_PIT_%d]\nPOS=%.1f, %.1f, %.1f\nROT=%.2f, -0.0, 0.0\n\n' % (k, c[i][0] + off * lat[i][0], c[i][1] + up, c[i][2] + off * lat[i][1], rot))
    cch = os.path.join(out, 'spline.cache')
    if os.path.exists(cch):
        os.remove(cch)
    return n



# ---------------------------------------------------------------------------------------------------------------------------
# Reference-scale collision meshes (BASELINE configs[4]).  The reference's default track
# content/tracks/driftplayground is 510 surfaces / 243 025 vertices / 112 411 triangles: 20 TRACK meshes (7 valid road,
# 13 grass-like with grip 0.8) and 490 separate WALL meshes -- a few long barriers (up to 1584 triangles, 145 m) and hundreds
# of metre-sized obstacles of ~100 triangles with centimetre edges -- in a 340 m x 210 m paddock with a 760 m closed spline
# of 498 points.  gen_playground builds a track of that shape from closed-form geometry, so it travels to machines that do not
# have the reference's content.
# ---------------------------------------------------------------------------------------------------------------------------
class _Lcg:
    """MSVC-style LCG: the same sequence on every machine and python version"""
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF

    def u(self):
        self.s = (self.s * 214013 + 2531011) & 0xFFFFFFFF
        return ((self.s >> 8) & 0xFFFFFF) / 16777216.0

    def r(self, a, b):
        return a + (b - a) * self.u()


def playground_ground(x, z):
    """height of the paddock: a tilted, gently rolling plane (the reference's lies between y = 60 and 77)"""
    return 68.0 + 0.012 * x + 0.9 * math.sin(x / 37.0) * math.cos(z / 29.0) + 0.5 * math.sin((x + z) / 61.0)


def playground_centreline(step=1.5):
    """closed circuit inside the paddock: a three-lobed ring, resampled every `step` metres"""
    m = 8000
    pts = []
    for k in range(m):
        th = 2.0 * math.pi * k / m
        r = 1.0 + 0.22 * math.cos(3.0 * th + 0.4) + 0.08 * math.sin(5.0 * th)
        pts.append((124.0 * r * math.cos(th) + 6.0, 76.0 * r * math.sin(th) - 3.0))
    cum = [0.0]
    for k in range(m):
        a, b = pts[k], pts[(k + 1) % m]
        cum.append(cum[-1] + math.hypot(b[0] - a[0], b[1] - a[1]))
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
        out.append((a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1])))
    return out


def _f32(v):
    return struct.unpack('<f', struct.pack('<f', v))[0]


def _grid_mesh(x0, x1, z0, z1, nx, nz, yfun):
    """heightfield patch, front faces up"""
    verts, idx = [], []
    for iz in range(nz + 1):
        z = z0 + (z1 - z0) * iz / nz
        for ix in range(nx + 1):
            x = x0 + (x1 - x0) * ix / nx
            verts.append((x, yfun(x, z), z))
    for iz in range(nz):
        for ix in range(nx):
            a = iz * (nx + 1) + ix; b = a + 1; c = a + nx + 1; d = c + 1
            idx += [a, c, d, a, d, b]          # (c-a)x(d-a): z then x => +y
    return verts, idx


def _revolve(cx, cz, y0, profile, segs, cap=True):
    """solid of revolution around the vertical through (cx, cz): profile = [(radius, height)], outward faces; a fan closes the top"""
    verts, idx = [], []
    for (r, hgt) in profile:
        for s in range(segs):
            a = 2.0 * math.pi * s / segs
            verts.append((cx + r * math.cos(a), y0 + hgt, cz + r * math.sin(a)))
    for k in range(len(profile) - 1):
        for s in range(segs):
            a = k * segs + s; b = k * segs + (s + 1) % segs; c = a + segs; d = b + segs
            idx += [a, c, b, b, c, d]
    if cap:
        top = len(verts)
        verts.append((cx, y0 + profile[-1][1], cz))
        base = (len(profile) - 1) * segs
        for s in range(segs):
            idx += [base + s, top, base + (s + 1) % segs]
    return verts, idx


def _extrude(path, section, yfun, closed=False):
    """barrier: the cross-section [(lateral offset, height)] swept along the polyline path [(x, z)] standing on the ground"""
    n = len(path)
    verts, idx = [], []
    m = len(section)
    for i in range(n):
        p0 = path[i - 1] if (closed or i > 0) else path[i]
        p1 = path[(i + 1) % n] if (closed or i + 1 < n) else path[i]
        fx, fz = p1[0] - p0[0], p1[1] - p0[1]
        fl = math.hypot(fx, fz)
        lx, lz = fz / fl, -fx / fl
        for (o, hgt) in section:
            x, z = path[i][0] + o * lx, path[i][1] + o * lz
            verts.append((x, yfun(path[i][0], path[i][1]) + hgt, z))
    segs = n if closed else n - 1
    for i in range(segs):
        for k in range(m - 1):
            a = i * m + k; b = a + 1; c = ((i + 1) % n) * m + k; d = c + 1
            idx += [a, b, c, b, d, c]
    return verts, idx


def _box_mesh(cx, cz, y0, hx, hy, hz, yaw, nx, ny, nz):
    """tessellated box (island / kerb block) standing on y0, rotated by yaw: five faces (no bottom)"""
    verts, idx = [], []
    cs, sn = math.cos(yaw), math.sin(yaw)

    def put(lx, ly, lz):
        verts.append((cx + cs * lx + sn * lz, y0 + ly, cz - sn * lx + cs * lz))

    def face(o, du, dv, nu, nv):
        b = len(verts)
        for j in range(nv + 1):
            for i in range(nu + 1):
                put(o[0] + du[0] * i / nu + dv[0] * j / nv, o[1] + du[1] * i / nu + dv[1] * j / nv, o[2] + du[2] * i / nu + dv[2] * j / nv)
        for j in range(nv):
            for i in range(nu):
                a = b + j * (nu + 1) + i; c = a + nu + 1
                idx.extend([a, c, c + 1, a, c + 1, a + 1])
    h2 = 2.0 * hy
    face((-hx, h2, -hz), (2 * hx, 0, 0), (0, 0, 2 * hz), nx, nz)            # top
    face((-hx, 0, -hz), (2 * hx, 0, 0), (0, h2, 0), nx, ny)                  # z- side
    face((hx, 0, hz), (-2 * hx, 0, 0), (0, h2, 0), nx, ny)                   # z+ side
    face((-hx, 0, hz), (0, 0, -2 * hz), (0, h2, 0), nz, ny)                  # x- side
    face((hx, 0, -hz), (0, 0, 2 * hz), (0, h2, 0), nz, ny)                   # x+ side
    return verts, idx


TYRE_STACK = [(0.40, 0.0), (0.46, 0.06), (0.46, 0.24), (0.40, 0.30), (0.46, 0.36), (0.46, 0.54), (0.40, 0.60), (0.46, 0.66), (0.46, 0.84), (0.40, 0.90)]
CONE = [(0.18, 0.0), (0.18, 0.03), (0.12, 0.05), (0.03, 0.50)]
BARREL = [(0.28, 0.0), (0.30, 0.05), (0.31, 0.45), (0.30, 0.85), (0.28, 0.90)]
JERSEY = [(0.30, 0.0), (0.30, 0.08), (0.12, 0.33), (0.08, 0.81), (-0.08, 0.81), (-0.12, 0.33), (-0.30, 0.08), (-0.30, 0.0)]


def gen_playground(out, step=1.5, side=6.0, seed=20260, stacks=300, cones=96, barrels=40, islands=24, ground_cell=2.4, trace_sides=False):
    """Synthetic paddock of the reference's driftplayground scale (see the block comment above): 13 grass-like ground strips +
    7 road chunks of the circuit ribbon (valid track, three kinds) = 20 TRACK surfaces; 4 perimeter walls, Jersey barriers in
    40 m runs along both sides of the circuit, tyre stacks (in front of the barriers, as chicanes on the road and as islands in
    the infield), cones, barrels and raised concrete islands inside the drivable area = ~490 separate WALL meshes.  Obstacles
    keep clear of the first 60 m of the racing line (the start pose).  Returns (surfaces, triangles, spline points)."""
    os.makedirs(out, exist_ok=True)
    g = playground_ground
    X0, X1, Z0, Z1 = -170.0, 170.0, -105.0, 105.0
    c = playground_centreline(step)
    n = len(c)
    lat = []
    for i in range(n):
        p0, p2 = c[i - 1], c[(i + 1) % n]
        fx, fz = p2[0] - p0[0], p2[1] - p0[1]
        fl = math.hypot(fx, fz)
        lat.append((fz / fl, -fx / fl))
    rng = _Lcg(seed)
    nsurf = 0; ntri = 0
    clear_pts = [c[i] for i in range(0, int(60.0 / step))] + [c[-i] for i in range(1, int(12.0 / step))]

    def near_start(x, z, d):
        return any((x - p[0]) ** 2 + (z - p[1]) ** 2 < d * d for p in clear_pts)
    placed = []

    def free(x, z, d):
        if not (X0 + 3.0 < x < X1 - 3.0 and Z0 + 3.0 < z < Z1 - 3.0) or near_start(x, z, 7.0):
            return False
        return all((x - p[0]) ** 2 + (z - p[1]) ** 2 >= (d + p[2]) ** 2 for p in placed)

    with open(os.path.join(out, 'surfaces.bin'), 'wb') as f:
        def emit(verts, idx, **kw):
            nonlocal nsurf, ntri
            assert len(verts) < 65536
            write_surface(f, [(_f32(v[0]), _f32(v[1]), _f32(v[2])) for v in verts], idx, **kw)
            nsurf += 1; ntri += len(idx) // 3
        # ground: 13 strips across x
        for k in range(13):
            xa = X0 + (X1 - X0) * k / 13.0; xb = X0 + (X1 - X0) * (k + 1) / 13.0
            v, ix = _grid_mesh(xa, xb, Z0, Z1, max(1, int(round((xb - xa) / ground_cell))), int(round((Z1 - Z0) / ground_cell)), g)
            emit(v, ix, sector=k, grip=0.8, valid=0, damping=0.0, dirt=0.0)
        # road: the circuit ribbon 3 cm above the ground, eight cells across, in 7 chunks of three kinds
        half = side + 1.0
        kinds = [dict(grip=0.97), dict(grip=0.98), dict(grip=0.95), dict(grip=0.97), dict(grip=0.98), dict(grip=0.95), dict(grip=0.97)]
        across = 8
        per = (n + 6) // 7
        for k in range(7):
            i0 = k * per; cnt = min(per, n - i0)
            verts, idx = [], []
            for j in range(cnt + 1):
                i = (i0 + j) % n
                for a in range(across + 1):
                    o = -half + 2.0 * half * a / across
                    x, z = c[i][0] + o * lat[i][0], c[i][1] + o * lat[i][1]
                    verts.append((x, g(x, z) + 0.03, z))
            for j in range(cnt):
                for a in range(across):
                    p = j * (across + 1) + a; q = p + 1; r = p + across + 1; s = r + 1
                    for tri in ((p, r, s), (p, s, q)):
                        v0, v1, v2 = verts[tri[0]], verts[tri[1]], verts[tri[2]]
                        ny = (v1[2] - v0[2]) * (v2[0] - v0[0]) - (v1[0] - v0[0]) * (v2[2] - v0[2])
                        idx.extend(tri if ny > 0 else (tri[0], tri[2], tri[1]))
            emit(verts, idx, sector=100 + k, valid=1, **kinds[k])
        # perimeter walls: 1 m panels, two bands high
        for (a, b) in (((X0, Z0), (X1, Z0)), ((X1, Z0), (X1, Z1)), ((X1, Z1), (X0, Z1)), ((X0, Z1), (X0, Z0))):
            L = math.hypot(b[0] - a[0], b[1] - a[1]); m = int(L)
            path = [(a[0] + (b[0] - a[0]) * i / m, a[1] + (b[1] - a[1]) * i / m) for i in range(m + 1)]
            v, ix = _extrude(path, [(0.0, -0.5), (0.0, 1.2), (0.0, 3.0)], g)
            emit(v, ix, sector=200, category=2, valid=0)
        # Jersey barriers: 40 m runs with 8 m gaps, 10.5 m either side of the centreline, one vertex ring per 0.75 m
        run = int(40.0 / step); gap = int(8.0 / step)
        for sgn in (-1.0, 1.0):
            i = int(70.0 / step) if sgn < 0 else int(82.0 / step)
            while i + run < n - int(14.0 / step):
                path = []
                for j in range(2 * run + 1):
                    t = i + 0.5 * j; i0 = int(t); fr = t - i0
                    pa, pb = c[i0 % n], c[(i0 + 1) % n]; la, lb = lat[i0 % n], lat[(i0 + 1) % n]
                    path.append((pa[0] + fr * (pb[0] - pa[0]) + sgn * (side + 4.5) * (la[0] + fr * (lb[0] - la[0])),
                                 pa[1] + fr * (pb[1] - pa[1]) + sgn * (side + 4.5) * (la[1] + fr * (lb[1] - la[1]))))
                v, ix = _extrude(path, JERSEY if sgn > 0 else JERSEY[::-1], g)
                emit(v, ix, sector=201, category=2, valid=0)
                for p in path[::4]:
                    placed.append((p[0], p[1], 0.5))
                i += run + gap
        # raised concrete islands inside the drivable area (kerb-high blocks, finely tessellated)
        k = 0
        while k < islands:
            i = int(rng.r(70.0 / step, n - 20.0 / step)); o = rng.r(-side + 1.0, side - 1.0) if (k % 3) else rng.r(-30.0, 30.0)
            x, z = c[i][0] + o * lat[i][0], c[i][1] + o * lat[i][1]
            if not free(x, z, 3.0):
                continue
            yaw = math.atan2(lat[i][1], lat[i][0]) + math.pi / 2 + rng.r(-0.3, 0.3)
            v, ix = _box_mesh(x, z, g(x, z) - 0.05, 0.5, 0.11, 2.0, yaw, 4, 2, 16)
            emit(v, ix, sector=202, category=2, valid=0)
            placed.append((x, z, 2.2)); k += 1
        # obstacles of revolution: tyre stacks (rows in front of the barriers, chicanes on the road, infield clusters), barrels, cones
        def drop(profile, segs, count, radius, where):
            k = 0; tries = 0
            while k < count and tries < 200000:
                tries += 1
                x, z = where(k)
                if not free(x, z, radius):
                    continue
                v, ix = _revolve(x, z, g(x, z) - 0.02, profile, segs)
                emit(v, ix, sector=203, category=2, valid=0)
                placed.append((x, z, radius)); k += 1
            assert k == count, (k, count)

        def along(lo, hi):
            def w(k):
                i = int(rng.r(62.0 / step, n - 16.0 / step)); o = rng.r(lo, hi) * (1.0 if rng.u() < 0.5 else -1.0)
                return c[i][0] + o * lat[i][0], c[i][1] + o * lat[i][1]
            return w

        def anywhere(k):
            return rng.r(X0 + 4.0, X1 - 4.0), rng.r(Z0 + 4.0, Z1 - 4.0)
        drop(TYRE_STACK, 8, stacks * 2 // 5, 0.55, along(side + 2.0, side + 3.6))      # in front of the barriers
        drop(TYRE_STACK, 8, stacks // 5, 0.55, along(1.5, side))                        # chicanes on the road
        drop(TYRE_STACK, 8, stacks - stacks * 2 // 5 - stacks // 5, 0.55, anywhere)    # infield / outfield
        drop(BARREL, 12, barrels, 0.4, along(0.5, side + 2.0))
        drop(CONE, 8, cones, 0.25, along(0.0, side))
    with open(os.path.join(out, 'spline.bin'), 'wb') as f:
        for i, p in enumerate(c):
            r = 1.5 * math.sin(i * step / 35.0)           # the racing line wanders across the road; sides follow
            x, z = p[0] + r * lat[i][0], p[1] + r * lat[i][1]
            f.write(struct.pack('<5f', x, g(x, z) + 0.03, z, side + r, side - r))
    with open(os.path.join(out, 'spline.ini'), 'w') as f:
        f.write('[SPLINE]\nCLOSED_LOOP=1\nTRACE_SIDES=%d\n' % (1 if trace_sides else 0))
    cch = os.path.join(out, 'spline.cache')
    if os.path.exists(cch):
        os.remove(cch)
    return nsurf, ntri, n


def read_spline_bin(path):
    """spline.bin: SlimTrackPoint = float[3] best + float[2] sides, no header (Sim/Track.cpp:97-150)"""
    raw = open(path, 'rb').read()
    n = len(raw) // 20
    return [struct.unpack_from('<5f', raw, 20 * i) for i in range(n)]


def gen_ribbon(out, points, closed=False, margin=4.0, points_per_surface=40, lead=3, walls=False, stagger=False):
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
    if stagger:     # cross-sections half way between the spline points: a vertical ray through a spline point then meets the inside
        #             of a triangle, not the shared edge of two (where float rounding can let it slip between them)
        pts = [tuple(0.5 * (pts[i][c] + pts[(i + 1) % len(pts)][c]) for c in range(5)) for i in range(len(pts) if closed else len(pts) - 1)]
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
            if walls:       # guard rails along both edges: one WALL mesh per side and road surface, 0.3 m below to 1.0 m above the edge
                for sgn in (-1.0, 1.0):
                    wv, wi = [], []
                    for k in range(cnt + 1):
                        e = edge((i0 + k) % n, sgn)
                        wv.append((e[0], e[1] - 0.3, e[2])); wv.append((e[0], e[1] + 1.0, e[2]))
                    for k in range(cnt):
                        a, b, cc, d = 2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 3
                        wi.extend((a, b, d, a, d, cc))
                    write_surface(f, wv, wi, sector=100000 + i0 // points_per_surface, category=2, valid=0)
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


def gen_nordring(out, points=13323, step=1.553, side=4.4, walls=True, **kw):
    """Synthetic stand-in for the reference's ks_nordschleife (13 323 spline points, 20.7 km, about 1.55 m apart,
    +-146 m of elevation, OPEN although its ends nearly meet): a large many-lobed ring with hills, resampled with uneven spacing, the
    racing line wandering across the road with the sides following; road = gen_ribbon around it, with guard rails (WALL surfaces)
    along both edges: 334 road surfaces + 668 rail meshes, ~80 k triangles."""
    os.makedirs(out, exist_ok=True)
    m = 120000
    raw = []
    for k in range(m):
        th = 2.0 * math.pi * k / m
        r = 2700.0 * (1.0 + 0.22 * math.sin(3.0 * th) + 0.09 * math.sin(5.0 * th + 1.0) + 0.035 * math.sin(17.0 * th + 0.3) + 0.012 * math.sin(41.0 * th))
        raw.append((r * math.cos(th), 90.0 * math.sin(2.0 * th) + 40.0 * math.sin(7.0 * th + 0.5) + 12.0 * math.sin(23.0 * th), r * math.sin(th)))
    cum = [0.0]
    for k in range(m):
        a, b = raw[k], raw[(k + 1) % m]
        cum.append(cum[-1] + math.sqrt(sum((b[i] - a[i]) ** 2 for i in range(3))))
    pts = []
    s, j = 0.0, 0
    for k in range(points):
        while j + 1 < m and cum[j + 1] < s:
            j += 1
        t = (s - cum[j]) / (cum[j + 1] - cum[j])
        a, b = raw[j], raw[(j + 1) % m]
        p = tuple(a[c] + t * (b[c] - a[c]) for c in range(3))
        fx, fz = b[0] - a[0], b[2] - a[2]
        fl = math.hypot(fx, fz)
        r = 1.6 * math.sin(s / 55.0)
        pts.append((p[0] + r * fz / fl, p[1], p[2] - r * fx / fl, side + r, side - r))
        s += step * (1.0 + 0.08 * math.sin(0.37 * k))
    assert s < cum[-1], 'the ring is shorter than the spline'
    with open(os.path.join(out, 'spline.bin'), 'wb') as f:
        for p in pts:
            f.write(struct.pack('<5f', *p))
    with open(os.path.join(out, 'spline.ini'), 'w') as f:
        f.write('[SPLINE]\nCLOSED_LOOP=0\nTRACE_SIDES=0\n')
    cch = os.path.join(out, 'spline.cache')
    if os.path.exists(cch):
        os.remove(cch)
    gen_ribbon(out, read_spline_bin(os.path.join(out, 'spline.bin')), closed=False, walls=walls, stagger=True, **kw)
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


GENERATORS = {'flat': gen_flat, 'touge': gen_touge, 'walled': gen_walled, 'hillclimb': gen_hillclimb, 'playground': gen_playground,
              'nordring': gen_nordring}


def make_base(base, tracks=('flat',)):
    """Create <base>/cfg/sim.ini (same keys/values as the reference's shipped cfg/sim.ini that the
    hot path reads: Sim/Simulator.cpp:62-75, Sim/Track.cpp:38-42,212-216, Car/Car.cpp:285-314) and the
    requested synthetic tracks under <base>/content/tracks/."""
    os.makedirs(os.path.join(base, 'cfg'), exist_ok=True)
    with open(os.path.join(base, 'cfg', 'sim.ini'), 'w') as f:
        f.write(SIM_INI)
    for t in tracks:
        GENERATORS[t](os.path.join(base, 'content', 'tracks', t))
    return base

if __name__ == '__main__':
    kind, out = sys.argv[1], sys.argv[2]
    GENERATORS[kind](out)
