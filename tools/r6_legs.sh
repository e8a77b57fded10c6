#!/bin/bash
# Diagnostic (GPU box): the contact-heavy legs + the headline for one library build.  usage: r6_legs.sh <label> <PDB_LIB or ""> [legs...]
LABEL=$1; LIB=$2; shift 2
if [ -n "$LIB" ]; then export PDB_LIB=$LIB; else unset PDB_LIB; fi
leg() { # name args...
  name=$1; shift
  python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-8s %-34s %7.2f M  %7.1f us/step  first pass %6.1f us' % ('$LABEL', '$name', j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"
}
for L in "$@"; do
case $L in
  headline) leg headline --steps 1500 --warmup 200 ;;
  pg_mlp) leg configs4_playground_16384_mlp --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200 ;;
  dp_mlp) leg configs4_driftplayground_16384_mlp --workload driftplayground --cars 16384 --policy mlp --episodes --teleport-mode 2 --steps 300 --warmup 50 --settle 200 ;;
  ep4096) leg episodes_4096 --workload touge --walls --cars 4096 --episodes --steps 600 --warmup 100 --settle 200 ;;
  ep4096rf) leg episodes_4096_reset_free --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200 ;;
  ep4096nc) leg episodes_4096_no_contacts --workload touge --walls --cars 4096 --policy feedback --no-body-contacts --steps 600 --warmup 100 --settle 200 ;;
  flat) leg configs1_flat_4096 --workload flat --steps 3000 --warmup 333 ;;
  flat16k) leg flat_16384 --workload flat --cars 16384 --steps 1500 --warmup 200 ;;
  nordring) leg configs4_nordring_16384_mlp --workload nordring --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200 ;;
esac
done
