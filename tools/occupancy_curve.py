"""Diagnostic: tick time against resident blocks per CU (flat plane, driving cars, one partition): 768 cars = one block of three
cars on each of the 256 CUs.  Flat time per added block = latency-bound; proportional growth = issue- or bandwidth-bound.
usage: occupancy_curve.py [cars ...]"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'projectd-core_amd'))
import pdbatch, parity_util as pu
P = pdbatch.packed_params(); trk = pdbatch.synthetic_track('flat')
sizes = [int(a) for a in sys.argv[1:]] or [768 * k for k in (1, 2, 3, 4, 5, 6, 7, 8, 12)]
for n in sizes:
    b = pdbatch.Batch(n, P, trk, 0, 1)
    b.upload_actions(pu.make_actions(n, 1234))
    b.step(700); b.sync()
    best = 1e9
    for rep in range(3):
        b.event_record(0); b.step(1000); b.event_record(1); b.sync()
        best = min(best, b.event_elapsed_ms() / 1000.0)
    print('blocks/CU %5.2f cars %5d  tick %.2f us  %.1f M env-steps/s' % (n / 768.0, n, best * 1e3, n / best / 1e3))
    b.close()
