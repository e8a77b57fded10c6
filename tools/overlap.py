#!/usr/bin/env python3
"""Diagnostic (GPU box): what runs beside what in a rocprofv3 --kernel-trace run (csv).  Over the last `frac` of the trace: per kernel class the summed duration, the wall
time with at least one kernel of the class running, and the wall time during which ONLY that class runs; the wall time with nothing running at all.
usage: overlap.py <dir with *_kernel_trace.csv> [frac=0.3]"""
import csv, glob, sys
d = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t_end = max(int(r['End_Timestamp']) for r in rows); t_beg = int(rows[0]['Start_Timestamp'])
lo = t_end - int((t_end - t_beg) * frac)
def cls(n):
    n = n.split('(')[0]
    if n.startswith('pdb_step_kernel'): return 'first'
    if n.startswith('pdb_contact_kernel'): return 'contact'
    if n.startswith('pdb_collide'): return 'collide'
    if n.startswith('pdb_resume'): return 'resume'
    return 'other'
ev = []
tot = {}; cnt = {}
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e <= lo: continue
    s = max(s, lo); c = cls(r['Kernel_Name'])
    tot[c] = tot.get(c, 0) + (e - s); cnt[c] = cnt.get(c, 0) + 1
    ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
act = {}; any_t = {}; only_t = {}; idle = 0; last = lo
for t, dlt, c in ev:
    dt = t - last
    if dt > 0:
        running = [k for k, v in act.items() if v > 0]
        if not running: idle += dt
        for k in running: any_t[k] = any_t.get(k, 0) + dt
        if len(running) == 1: only_t[running[0]] = only_t.get(running[0], 0) + dt
    act[c] = act.get(c, 0) + dlt; last = t
wall = t_end - lo
print('window %.1f ms; nothing running %.1f %%' % (wall / 1e6, 100.0 * idle / wall))
for c in sorted(tot):
    print('%-8s launches %6d  avg %8.1f us  summed %6.1f %% of wall  some running %5.1f %%  only this class %5.1f %%' % (c, cnt[c], tot[c] / cnt[c] / 1e3, 100.0 * tot[c] / wall, 100.0 * any_t.get(c, 0) / wall, 100.0 * only_t.get(c, 0) / wall))
