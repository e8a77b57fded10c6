#!/bin/bash
# Diagnostic (GPU box): waves on a car's narrow phase in the collide kernel (development builds of the 33-row class: tools/variants/libpdbatch_nw{4,8}.so)
leg() { label=$1; lib=$2; shift 2; PDB_LIB=../tools/variants/$lib python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-44s %7.2f M  %7.1f us/step' % ('$label', j['value']/1e6, j['ms_per_step']*1e3)); break
"; }
for r in 1 2; do for v in nw4 nw2 nw1 nw4c1 nw2c1; do
  leg "$v playground 16384 mlp" libpdbatch_$v.so --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200
  leg "$v driftplayground 16384 mlp env" libpdbatch_$v.so --workload driftplayground --cars 16384 --policy mlp --episodes --teleport-mode 2 --steps 300 --warmup 50 --settle 200
  leg "$v walled road 4096 reset-free" libpdbatch_$v.so --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200
done; done
