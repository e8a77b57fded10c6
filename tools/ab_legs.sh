#!/bin/bash
# Diagnostic (GPU box): two playground legs for library variants (PDB_LIB) against the in-tree one.  usage: ab_legs.sh variant.so ...
B="--cars 16384 --steps 100 --warmup 20 --settle 200 --no-cpu-baseline --no-extra --workload playground"
line() { lib=$1; shift; if [ -n "$lib" ]; then export PDB_LIB=$lib; else unset PDB_LIB; fi
  timeout 200 python3 bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ${lib:-in-tree} $*: %.2f M env-steps/s, contact-pass cars %s' % (d['value']/1e6, d.get('contact_pass_cars')))" || echo "  ${lib:-in-tree} $*: FAILED"; }
for v in "$@" ""; do line "$v" --episodes; line "$v" --policy mlp; done
