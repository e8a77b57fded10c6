#!/bin/bash
# Diagnostic (GPU box): where a tick goes on the reference-scale meshes -- kernel stats of the playground / nordring legs.
# Usage: bash tools/scale_probe.sh   (outputs under gpurun_out/r3scale/); every command under its own timeout
set -u
OUT=gpurun_out/r3scale
mkdir -p $OUT
export TMPDIR=/tmp
B="--cars 16384 --steps 100 --warmup 20 --settle 200 --no-cpu-baseline --no-extra"
stats() {  # name, bench args
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1 -o run -- python3 bench.py ${@:2} $B > $OUT/stats_$1.log 2>&1
  python3 - <<PY
import csv,glob
for f in glob.glob('$OUT/stats_$1/**/run_kernel_stats.csv', recursive=True) or glob.glob('$OUT/stats_$1/run_kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    print('== $1')
    for r in rows[:3]: print('  %-50s calls %6s avg %10.1f us  total %5.1f%%' % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
    break
PY
}
line() { timeout 200 python3 bench.py ${@:2} $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1: %.2f M env-steps/s, %.1f us per partition tick, contact-pass cars %s%s' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d.get('contact_pass_cars'), (', ends/tick %.2f' % d['episode_ends_per_tick']) if 'episode_ends_per_tick' in d else ''))" || echo "  $1: FAILED"; }
if [ "${1:-all}" != "lines" ]; then
stats pg_episodes --workload playground --episodes
stats pg_mlp --workload playground --policy mlp
fi
line "playground episodes" --workload playground --episodes
line "playground mlp" --workload playground --policy mlp
line "playground feedback" --workload playground --policy feedback
line "nordring mlp" --workload nordring --policy mlp
line "nordring feedback" --workload nordring --policy feedback
line "nordring episodes" --workload nordring --episodes
line "touge walls mlp" --workload touge --walls --policy mlp
line "touge walls episodes" --workload touge --walls --episodes
line "touge walls host (pipelined)" --workload touge --walls --policy host
line "touge walls host_sync" --workload touge --walls --policy host_sync
