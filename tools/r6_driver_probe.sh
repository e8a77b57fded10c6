#!/bin/bash
# Diagnostic (GPU box): what the driver's 20-step regions pay for the sampled launches' HIP events, and two partitions against three in that regime
leg() { # label args...
  label=$1; shift
  python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-34s %7.2f M  %7.1f us/step  first pass %6.1f us' % ('$label', j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"
}
for r in 1 2; do
  leg "sample every 4 (default)"
  leg "sample every 10" --sample-every 10
  leg "no sampling" --sample-every 0
  leg "two partitions" --partitions 2
done
