#!/bin/bash
# Diagnostic (GPU box): in-run A/B of two library builds on the headline workload (configs[2] as worded) and on configs[1].  usage: r5_ab.sh <out dir> <variant .so relative to the package dir> [rounds]
OUT=$1; VAR=$2; N=${3:-2}; mkdir -p $OUT
one() { # label lib args...
  label=$1; lib=$2; shift 2
  if [ -n "$lib" ]; then export PDB_LIB=$lib; else unset PDB_LIB; fi
  python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-10s %-40s %.2f M  %.1f us/step  first pass %.1f us' % ('$label', '$*'[:40], j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"
}
for i in $(seq $N); do
  one base "$VAR" --steps 1500 --warmup 200
  one new "" --steps 1500 --warmup 200
  one base "$VAR" --workload flat --steps 3000 --warmup 333
  one new "" --workload flat --steps 3000 --warmup 333
done 2>&1 | tee $OUT/ab.txt
