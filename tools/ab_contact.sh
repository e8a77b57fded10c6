#!/bin/bash
# Diagnostic (GPU box): the contact-heavy legs with library variants (PDB_LIB, paths relative to projectd-core_amd/) against the in-tree library, alternating, same box.
# usage: ab_contact.sh [variant.so ...]
line() { lib=$1; shift; label=$1; shift; if [ -n "$lib" ]; then export PDB_LIB=$lib; else unset PDB_LIB; fi
  timeout 300 python3 bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $label: %.2f M env-steps/s, %.1f us per tick, contact pass cars %s' % (d['value']/1e6, d['ms_per_step']*1e3, d.get('contact_pass_cars')))" || echo "  $label: FAILED"; }
S="--steps 300 --warmup 50 --settle 200"
for v in "$@" ""; do n=${v:-in-tree}
  line "$v" "$n, playground 16384 mlp" --workload playground --cars 16384 --policy mlp $S
  line "$v" "$n, nordring 16384 mlp" --workload nordring --cars 16384 --policy mlp $S
  line "$v" "$n, driftplayground 16384 mlp episodes" --workload driftplayground --cars 16384 --policy mlp --episodes --teleport-mode 2 $S
  line "$v" "$n, walled road 4096 reset-free" --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200
  line "$v" "$n, walled road 4096 mlp" --workload touge --walls --cars 4096 --policy mlp --steps 600 --warmup 100 --settle 200
  line "$v" "$n, walled road 16384 mlp" --workload touge --walls --cars 16384 --policy mlp $S
done
