#!/bin/bash
# Diagnostic (GPU box): the bench line with library variants (PDB_LIB: paths relative to projectd-core_amd/, e.g. ../tools/variants/libpdbatch_r3.so = the round-3
# final library built from commit ed498b4) against the in-tree one, alternating, same box.  usage: ab_headline.sh [variant.so ...]
H="--no-cpu-baseline --no-extra --steps 3000 --warmup 333"
line() { lib=$1; shift; label=$1; shift; if [ -n "$lib" ]; then export PDB_LIB=$lib; else unset PDB_LIB; fi
  timeout 200 python3 bench.py $H "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $label: %.2f M env-steps/s, %.1f us per partition tick' % (d['value']/1e6, d['roofline']['kernel_avg_us']))" || echo "  $label: FAILED"; }
for rep in 1 2; do
  for v in "$@"; do line $v "$v, 4096 flat"; done
  line "" "in-tree, 4096 flat"
done
for v in "$@"; do line $v "$v, 16384 flat" --cars 16384; done
line "" "in-tree, 16384 flat" --cars 16384
line "" "in-tree, 4096 no body contacts" --no-body-contacts
