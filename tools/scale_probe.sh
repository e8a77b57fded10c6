#!/bin/bash
# Diagnostic (GPU box): where a tick goes on the reference-scale meshes -- kernel stats of the playground / nordring legs, and
# the contact pass's grid size.  Usage: bash tools/scale_probe.sh   (outputs under gpurun_out/r3scale/)
set -u
OUT=gpurun_out/r3scale
mkdir -p $OUT
export TMPDIR=/tmp
B="--cars 16384 --steps 100 --warmup 20 --settle 200 --no-cpu-baseline --no-extra"
for pol in mlp feedback; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pg_$pol -o run -- python3 bench.py --workload playground --policy $pol $B > $OUT/stats_pg_$pol.log 2>&1
  python3 - <<PY
import csv,glob
for f in glob.glob('$OUT/stats_pg_$pol/**/run_kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    print('== playground $pol')
    for r in rows[:6]: print('  %-60s calls %6s avg %10.1f us  total %5.1f%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
done
for g in 32 128 512 1536; do
  echo "== PDB_CONTACT_GRID=$g"
  PDB_CONTACT_GRID=$g python3 bench.py --workload playground --policy mlp $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  playground mlp: %.2f M env-steps/s, %.1f us per partition tick' % (d['value']/1e6, d['roofline']['kernel_avg_us']))"
  PDB_CONTACT_GRID=$g python3 bench.py --workload playground --policy feedback $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  playground feedback: %.2f M env-steps/s, %.1f us per partition tick' % (d['value']/1e6, d['roofline']['kernel_avg_us']))"
done
PDB_CONTACT_GRID=512 python3 bench.py --workload playground --episodes $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  playground episodes grid 512: %.2f M env-steps/s, ends/tick %.2f' % (d['value']/1e6, d.get('episode_ends_per_tick',-1)))"
PDB_CONTACT_GRID=512 python3 bench.py --workload nordring --policy mlp $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  nordring mlp grid 512: %.2f M env-steps/s' % (d['value']/1e6))"
PDB_CONTACT_GRID=512 python3 bench.py --workload nordring --policy feedback $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  nordring feedback grid 512: %.2f M env-steps/s' % (d['value']/1e6))"
PDB_CONTACT_GRID=512 python3 bench.py --workload touge --walls --policy mlp $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  touge walls mlp grid 512: %.2f M env-steps/s' % (d['value']/1e6))"
python3 bench.py --workload touge --walls --policy mlp $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  touge walls mlp grid 32: %.2f M env-steps/s' % (d['value']/1e6))"
