#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE (build container only).  Assembles the base directory the
reference-TU harness runs in: oracle/_ref/base/{cfg/sim.ini, content/cars/<model>/data, content/tracks/*}.
Car data and sim.ini are copied from /root/reference into oracle/_ref (git-ignored build output);
nothing from the reference enters the repository history."""
import os, shutil, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, '..', 'projectd-core_amd'))
import synthetic_tracks as gen_track
REF = '/root/reference'
def main():
    base = os.path.join(here, '_ref', 'base')
    os.makedirs(os.path.join(base, 'cfg'), exist_ok=True)
    shutil.copy(os.path.join(REF, 'cfg', 'sim.ini'), os.path.join(base, 'cfg', 'sim.ini'))
    for model in os.listdir(os.path.join(REF, 'content', 'cars')):
        dst = os.path.join(base, 'content', 'cars', model, 'data')
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        shutil.copytree(os.path.join(REF, 'content', 'cars', model, 'data'), dst)
        os.system('chmod -R u+w "%s"' % dst)
    gen_track.gen_flat(os.path.join(base, 'content', 'tracks', 'flat'))
    gen_track.gen_touge(os.path.join(base, 'content', 'tracks', 'touge'))
    for trk in ('driftplayground',):
        dst = os.path.join(base, 'content', 'tracks', trk)
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        shutil.copytree(os.path.join(REF, 'content', 'tracks', trk), dst)
        os.system('chmod -R u+w "%s"' % dst)
    print(base)
if __name__ == '__main__':
    main()
