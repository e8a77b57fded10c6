#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE (build container only).  Regenerates tests/golden/*.npz from the
reference-TU harness: build oracle/_ref/refharness (oracle/refharness/Makefile), assemble the base
directory (make_base.py: synthetic tracks, derived cars, copies of the shipped tracks that come with their mesh), run every
scripted scenario (oracle/scenarios.h) and store the probe records.  A fixture is data: actions in, reference-computed values out."""
import os, subprocess, sys, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import probe_io
def main():
    subprocess.check_call(['make', '-C', os.path.join(here, 'refharness')])
    subprocess.check_call([sys.executable, os.path.join(here, 'make_base.py')])
    out = os.path.join(here, '_ref', 'out'); os.makedirs(out, exist_ok=True)
    subprocess.check_call([os.path.join(here, '_ref', 'refharness'), os.path.join(here, '_ref', 'base'), 'all', out])
    gold = os.path.join(here, '..', 'tests', 'golden'); os.makedirs(gold, exist_ok=True)
    for f in sorted(os.listdir(out)):
        if not f.endswith('.bin'): continue
        p = probe_io.load(os.path.join(out, f))
        d = p['data']
        # float32 where lossless (most fields), float64 otherwise
        f32 = d.astype(np.float32)
        lossless = (f32.astype(np.float64) == d).all(axis=0)
        np.savez_compressed(os.path.join(gold, f[:-4] + '.npz'), names=np.array(p['names']), ticks=p['ticks'], actions=p['actions'],
                            f32_cols=np.where(lossless)[0].astype(np.int32), f32=f32[:, lossless],
                            f64_cols=np.where(~lossless)[0].astype(np.int32), f64=d[:, ~lossless])
        print(f, d.shape, 'f64 cols', int((~lossless).sum()))
if __name__ == '__main__':
    main()
