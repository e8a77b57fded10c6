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

def write_surface(f, verts, indices, sector=0, category=1, grip=1.0, valid=1):
    hdr = struct.pack('<5I9f2B', MAGIC, len(verts), len(indices), sector, category,
                      grip, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, valid, 0)
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
        {'flat': gen_flat}[t](os.path.join(base, 'content', 'tracks', t))
    return base

if __name__ == '__main__':
    kind, out = sys.argv[1], sys.argv[2]
    {'flat': gen_flat}[kind](out)
