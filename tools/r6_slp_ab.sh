#!/bin/bash
# Diagnostic (GPU box): the device code compiled without the SLP vectorizer (-fno-slp-vectorize: no v_pk_* pairs, a tenth fewer instructions) against the default; development builds
leg() { label=$1; lib=$2; shift 2; PDB_LIB=../tools/variants/$lib python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-40s %7.2f M  %7.1f us/step  first pass %6.1f us' % ('$label', j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"; }
for r in 1 2; do for v in ${VARIANTS:-base noslp}; do
  leg "$v headline 1500" libpdbatch_$v.so --steps 1500 --warmup 200
  leg "$v headline driver-style" libpdbatch_$v.so --steps 20 --warmup 5
  leg "$v flat 4096" libpdbatch_$v.so --workload flat --steps 3000 --warmup 333
  leg "$v flat 16384" libpdbatch_$v.so --workload flat --cars 16384 --steps 1500 --warmup 200
  leg "$v playground 16384 mlp" libpdbatch_$v.so --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200
  leg "$v walled road 4096 env loop" libpdbatch_$v.so --workload touge --walls --cars 4096 --episodes --steps 600 --warmup 100 --settle 200
done; done
