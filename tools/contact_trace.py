#!/usr/bin/env python3
"""Diagnostic (GPU box): the longest pdb_contact_kernel launches of a rocprofv3 --kernel-trace run, with their place in the launch sequence and
their grid -- which launch met a queue its grid was not sized for.  usage: contact_trace.py <dir with *_kernel_trace.csv> [threshold_us]"""
import csv, glob, sys
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
fs = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']) if rows else 0
seq = {}
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'pdb_' not in n:
        continue
    k = n.split('(')[0]
    seq[k] = seq.get(k, 0) + 1
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'contact' in k:
        out.append((dur, seq[k], (int(r['Start_Timestamp']) - t0) / 1e6, int(r.get('Grid_Size_X', r.get('Grid_Size', 0))), int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))), k))
print('%d contact-pass launches; above %.0f us: %d; max %.1f us' % (len(out), thr, sum(1 for o in out if o[0] > thr), max(o[0] for o in out) if out else 0))
for dur, i, t, g, wg, k in sorted(out, reverse=True)[:12]:
    print('  %9.1f us  launch #%-6d at %8.1f ms  grid %5d workgroups  %s' % (dur, i, t, g // max(wg, 1), k))
