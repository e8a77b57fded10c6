#!/bin/bash
# Diagnostic (GPU box): the MLP stand-in's four launches replayed from a captured graph at 16384 cars (--graph-max-cars) against plain launches
leg() { label=$1; shift; python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-44s %7.2f M  %7.1f us/step' % ('$label', j['value']/1e6, j['ms_per_step']*1e3)); break
"; }
for r in 1 2; do
  leg "playground mlp, plain launches" --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200
  leg "playground mlp, policy from a graph" --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200 --graph-max-cars 100000
  leg "nordring mlp, plain launches" --workload nordring --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200
  leg "nordring mlp, policy from a graph" --workload nordring --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200 --graph-max-cars 100000
done
