#!/usr/bin/env python3
"""Build container only: pack the reference's shipped tracks into the product's own track-blob format, compressed
(projectd-core_amd/data/tracks/<name>.pdtrack.z), so that they travel to machines without the reference's content/ directory
the way the cars' .pdcar blocks do.  The four tracks that ship with their mesh (driftplayground, ebisu_touge, yamanashi_short,
euphoria_hillside_park) are built from it (Sim/Track.cpp:97-272: surfaces.bin + spline.bin + spline.cache); ek_akina and
ks_nordschleife ship with their spline only (surfaces.bin is a missing blob), so the road is a ribbon generated around the
spline (synthetic_tracks.ribbon_track_from).  Output is git-ignored build output, like the .so files."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdb_ctypes as pc
import synthetic_tracks
REF = '/root/reference'
MESH_TRACKS = ('driftplayground', 'ebisu_touge', 'yamanashi_short', 'euphoria_hillside_park')
RIBBON_TRACKS = ('ek_akina', 'ks_nordschleife')
WALLED_RIBBONS = ('ks_nordschleife_walls',)   # the same ribbon with guard rails along both edges (WALL surfaces): BASELINE configs[4]'s collision mesh


def build_blob(lib, name, scratch):
    if name in RIBBON_TRACKS or name in WALLED_RIBBONS:
        src = name[:-len('_walls')] if name in WALLED_RIBBONS else name
        synthetic_tracks.ribbon_track_from(os.path.join(REF, 'content', 'tracks', src), os.path.join(scratch, 'content', 'tracks', name),
                                           walls=name in WALLED_RIBBONS)
        return pc.build_track(lib, scratch, name)
    return pc.build_track(lib, REF, name)


def main(force=False):
    lib = pc.load_product(host_only=True)
    os.makedirs(pc.TRACK_PACK_DIR, exist_ok=True)
    scratch = synthetic_tracks.make_base(tempfile.mkdtemp(prefix='pdb_pack_'), tracks=())   # cfg/sim.ini: the reference's shipped values
    for name in MESH_TRACKS + RIBBON_TRACKS + WALLED_RIBBONS:
        path = pc.track_pack_path(name)
        if os.path.exists(path) and not force:
            continue
        blob = build_blob(lib, name, scratch)
        pc.write_track_pack(path, blob)
        print('%-26s %6.1f MB blob -> %5.1f MB pack' % (name, len(blob) / 1e6, os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    main(force='--force' in sys.argv)
