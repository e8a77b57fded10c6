#!/bin/bash
# Diagnostic (GPU box): the kernel classes compiled for exactly 26 / 38 rows against the row-guarded kernels of the 33- / 40-row classes (PDB_NO_EXACT_CLASSES=1), same library
leg() { # label car args...
  label=$1; car=$2; shift 2
  PDB_BENCH_CAR=$car python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-10s %-32s %7.2f M  %7.1f us/step  first pass %6.1f us' % ('$label', '$car', j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"
}
for r in 1 2; do for v in exact guarded; do
  if [ $v = guarded ]; then export PDB_NO_EXACT_CLASSES=1; else unset PDB_NO_EXACT_CLASSES; fi
  for car in ks_mazda_rx7_tuned ks_toyota_supra_mkiv_drift dthwsh_mazda_rx7_fc3s_sr20 gravygarage_street_ae86_readie; do
    leg $v $car --workload flat --cars 16384 --steps 1000 --warmup 200
  done
  leg $v ks_mazda_rx7_tuned --workload touge --cars 16384 --steps 300 --warmup 50 --settle 200
  leg $v dthwsh_mazda_rx7_fc3s_sr20 --workload touge --cars 16384 --steps 300 --warmup 50 --settle 200
done; done
