#!/usr/bin/env python3
"""Diagnostic (GPU box): a stretch of a rocprofv3 --kernel-trace run as a timeline -- start, duration, queue, kernel, grid -- to see what runs beside what.
usage: timeline.py <dir with *_kernel_trace.csv> [first kernel index] [count]"""
import csv, glob, sys
d = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else -400; cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
if first < 0:
    first = max(0, len(rows) + first)
sel = rows[first:first + cnt]
t0 = int(sel[0]['Start_Timestamp'])
qs = {}
for r in sel:
    q = r.get('Queue_Id', r.get('Stream_Id', '?'))
    qs.setdefault(q, len(qs))
    n = r['Kernel_Name'].split('(')[0]
    n = {'pdb_step_kernel': 'FIRST', 'pdb_contact_kernel': 'CONTACT'}.get(n, n[:28])
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))))
    print('%9.1f %9.1f  %7.1f us  q%-2d %s%-28s grid %5d' % (s, e, e - s, qs[q], '    ' * qs[q], n, g))
