#!/bin/bash
# Diagnostic (GPU box): the device law (pdb_set_law: the linear laws evaluated by the tick's own launches) against the torch launch behind every tick, same library,
# alternating repeats.  usage: r6_law_ab.sh [repeats]
R=${1:-2}
leg() { # label name args...
  label=$1; name=$2; shift 2
  python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-12s %-30s %7.2f M  %7.1f us/step  first pass %6.1f us' % ('$label', '$name', j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"
}
for r in $(seq $R); do
  for v in law torch; do
    X=""; [ $v = torch ] && X="--no-device-law"
    D=""; [ $v = law ] && D="--device-law"; leg $v headline_1500 --steps 1500 --warmup 200 $X $D
    leg $v headline_driver --steps 20 --warmup 5 $X $D
    leg $v episodes_4096 --workload touge --walls --cars 4096 --episodes --steps 600 --warmup 100 --settle 200 $X
    leg $v episodes_4096_reset_free --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200 $X
    leg $v configs2_16384_touge --workload touge --cars 16384 --steps 300 --warmup 50 --settle 200 $X
    leg $v nordring_16384_feedback --workload nordring --cars 16384 --steps 300 --warmup 50 --settle 200 $X
  done
done
