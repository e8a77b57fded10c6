#!/bin/bash
# Diagnostic (GPU box): free-running partitions 2 / 3 / 4 on the legs whose loop holds no torch launch any more (the device law), with 4 and 8 hardware queues
leg() { # label name args...
  label=$1; name=$2; shift 2
  python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-22s %-30s %7.2f M  %7.1f us/step' % ('$label', '$name', j['value']/1e6, j['ms_per_step']*1e3)); break
"
}
for q in 4 8; do for p in 2 3 4; do
  export GPU_MAX_HW_QUEUES=$q
  leg "queues $q parts $p" episodes_4096_reset_free --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200 --partitions $p
  leg "queues $q parts $p" episodes_4096 --workload touge --walls --cars 4096 --episodes --steps 600 --warmup 100 --settle 200 --partitions $p
  leg "queues $q parts $p" playground_16384_feedback --workload playground --cars 16384 --policy feedback --steps 300 --warmup 50 --settle 200 --partitions $p
  [ $q = 4 ] && leg "queues $q parts $p" headline_1500 --steps 1500 --warmup 200 --partitions $p
done; done
