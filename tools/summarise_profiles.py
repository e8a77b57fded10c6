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


# issue model: tools/issue_rate.hip's own table + the SQ counters of the same launches
import re
im = {'source': 'tools/issue_rate.hip on the GPU box (s_memtime; every CU busy, independent instruction streams)', 'cycles_per_inst_per_simd': {}, 'counters_per_instruction': {}}
ir = os.path.join(src, 'issue_rate.txt')
if os.path.exists(ir):
    txt = open(ir).read()
    open(os.path.join(dst, tag + '_issue_rate.txt'), 'w').write(txt)
    for m in re.finditer(r'^(.+?)\s+waves/SIMD (\d+): ([0-9.]+) cycles per wave-instruction per SIMD \(wave sees ([0-9.]+)\)', txt, re.M):
        im['cycles_per_inst_per_simd'].setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
    acc = {}
    for r in rows('pmc_issue/**/*counter_collection.csv'):
        if 'stream_kernel' not in r['Kernel_Name']:
            continue
        k = (r['Kernel_Name'].split('(')[0], r['Grid_Size'], r['Dispatch_Id'])
        acc.setdefault(k, {})[r['Counter_Name']] = acc.setdefault(k, {}).get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for (kn, grid, did), v in acc.items():
        if v.get('SQ_INSTS_VALU', 0) < 1e7:
            continue   # the short warm-up launch
        wps = int(grid) // 256 // 256
        im['counters_per_instruction'].setdefault(kn, {})[str(wps)] = {'SQ_ACTIVE_INST_VALU_quads_per_VALU_inst': v.get('SQ_ACTIVE_INST_VALU', 0) / v['SQ_INSTS_VALU'],
                                                                        'SQ_WAVE_CYCLES_quads_per_VALU_inst_per_wave': v.get('SQ_WAVE_CYCLES', 0) / v['SQ_INSTS_VALU']}
    json.dump(im, open(os.path.join(dst, tag + '_issue_model.json'), 'w'), indent=1)
summary = {'tag': tag, 'kernel': KERNEL, 'workload_key': (sys.argv[2] if len(sys.argv) > 2 else 'ek_akina')}
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
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq1', 'pmc_sq2', 'pmc_sq3'):
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
        if 'SQ_INSTS_VALU' in cnt and 'rocprof_avg_us' in summary:
            # VALU issue occupancy, priced with MEASURED issue costs (tools/issue_rate.hip, <tag>_issue_model.json): at this kernel's
            # 6 waves per SIMD a SIMD issues one fp32 wave-instruction per ISSUE_F32 cycles and one fp64 per ISSUE_F64.
            # SQ_ACTIVE_INST_VALU is NOT a busy time: the same run shows it is exactly 1 quad-cycle (4 cycles) per VALU
            # instruction at every occupancy and for fp32 and fp64 alike (2 for v_rcp_f32), i.e. an instruction count.
            # The counter passes serialise the launches, so counts are per launch; `conc` launches share the 1024 SIMDs in the
            # timed run, each taking rocprof_avg_us; clock = the in-kernel clock of the bench kernel's own stamps (~2.1 GHz).
            issue = {}
            try:
                issue = json.load(open(os.path.join(dst, tag + '_issue_model.json')))
            except Exception:
                pass
            c32 = issue.get('cycles_per_inst_per_simd', {}).get('v_fma_f32', {}).get('6', 1.46)
            c64 = issue.get('cycles_per_inst_per_simd', {}).get('v_fma_f64', {}).get('6', 2.34)
            clk = 2.1e9
            simd_cycles = summary['rocprof_avg_us'] * 1e-6 * clk * 1024
            summary['concurrent_launches'] = conc
            summary['valu_issue_busy_frac'] = conc * cnt['SQ_INSTS_VALU'] * c32 / simd_cycles
            summary['valu_issue_busy_frac_upper'] = conc * cnt['SQ_INSTS_VALU'] * c64 / simd_cycles
            summary['valu_note'] = ('share of all SIMD issue cycles taken by this kernel\'s VALU instructions with %d launch(es) in flight, if every instruction cost what an '
                                    'independent v_fma_f32 costs at 6 waves per SIMD (%.2f cycles, measured); _upper: if every one cost a v_fma_f64 (%.2f)' % (conc, c32, c64))
            if waves:
                pw = summary['per_wave']
                tot = pw.get('SQ_WAVE_CYCLES', 0) or 1.0
                summary['wave_time_split'] = {'issuing (SQ_ACTIVE_INST_ANY)': pw.get('SQ_ACTIVE_INST_ANY', 0) / tot, 'parked on s_waitcnt / barrier / sleep (SQ_WAIT_ANY)': pw.get('SQ_WAIT_ANY', 0) / tot,
                                              'issue stall (SQ_WAIT_INST_ANY)': pw.get('SQ_WAIT_INST_ANY', 0) / tot}
json.dump(summary, open(os.path.join(dst, tag + '_pmc.json'), 'w'), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != 'bench'}, indent=1))
