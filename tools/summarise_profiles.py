#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into the small files committed under profiles/:
<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc.json (per-launch / per-wave counters of
pdb_step_kernel, HBM-side traffic with the gfx950 FETCH_SIZE correction), <tag>_bench.json (the bench line)."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join('gpurun_out', tag)
dst = 'profiles'
os.makedirs(dst, exist_ok=True)
KERNEL = 'pdb_step_kernel'


def rows(pattern):
    for f in glob.glob(os.path.join(src, pattern), recursive=True):
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                yield r


summary = {'tag': tag, 'kernel': KERNEL}
# kernel stats
ks = [r for r in rows('stats/**/*kernel_stats.csv')]
if ks:
    with open(os.path.join(dst, tag + '_kernel_stats.csv'), 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=list(ks[0].keys())); w.writeheader(); w.writerows(ks)
    for r in ks:
        if KERNEL in r.get('Name', ''):
            summary['rocprof_avg_us'] = float(r['AverageNs']) / 1000.0
            summary['rocprof_calls'] = int(r['Calls'])
# counters: average per dispatch of the step kernel
cnt = {}
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq1', 'pmc_sq2'):
    acc = {}
    for r in rows(d + '/**/*counter_collection.csv'):
        if KERNEL not in r['Kernel_Name']:
            continue
        a = acc.setdefault(r['Counter_Name'], [0.0, 0])
        a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (s, n) in acc.items():
        cnt[k] = s / n
        cnt.setdefault('_grid', None)
summary['per_launch'] = {k: v for k, v in cnt.items() if not k.startswith('_')}
waves = cnt.get('SQ_WAVES')
if waves:
    summary['per_wave'] = {k: v / waves for k, v in cnt.items() if k.startswith('SQ_') and k != 'SQ_WAVES'}
    pw = summary['per_wave']
if 'FETCH_SIZE' in cnt and 'WRITE_SIZE' in cnt:
    # units: KiB (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); gfx950: FETCH_SIZE reads 1/2 of wide coalesced
    # streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled.  Memory-side (fabric) requests incl. Infinity-Cache hits.
    summary['traffic_bytes_per_launch'] = (2.0 * cnt['FETCH_SIZE'] + cnt['WRITE_SIZE']) * 1024.0
    summary['traffic_note'] = 'memory-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; the 9 MB state is Infinity-Cache resident between ticks'
b = os.path.join(src, 'bench.json')
if os.path.exists(b):
    line = [l for l in open(b).read().splitlines() if l.startswith('{')]
    if line:
        bj = json.loads(line[-1]); summary['bench'] = bj
        json.dump(bj, open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)
        conc = bj.get('roofline', {}).get('concurrent_launches', 1) or 1
        if 'SQ_ACTIVE_INST_VALU' in cnt and 'rocprof_avg_us' in summary:
            # VALU issue roofline.  SQ_ACTIVE_INST_VALU = quad-cycles a SIMD spends issuing VALU work, summed over the launch's waves
            # (the counter passes serialise the launches, so this is per launch); `conc` launches share the 1024 SIMDs in the
            # timed run, each taking rocprof_avg_us; 2.4 GHz nominal shader clock.
            summary['concurrent_launches'] = conc
            summary['valu_busy_frac_approx'] = conc * 4.0 * cnt['SQ_ACTIVE_INST_VALU'] / (summary['rocprof_avg_us'] * 1e-6 * 2.4e9 * 1024)
            summary['valu_note'] = 'fraction of all SIMD issue cycles spent on VALU instructions with %d launch(es) in flight: the resource that bounds this kernel' % conc
json.dump(summary, open(os.path.join(dst, tag + '_pmc.json'), 'w'), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != 'bench'}, indent=1))
