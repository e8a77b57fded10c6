#!/bin/bash
# Diagnostic (GPU box): the bench line with the round-2 final library (projectd-core_amd/libpdbatch_r2.so, built from commit ed498b4)
# and other variants (PDB_LIB) against the in-tree one, alternating, same box.  Every command under its own timeout.
H="--no-cpu-baseline --no-extra --steps 3000 --warmup 333"
line() { lib=$1; shift; label=$1; shift; if [ -n "$lib" ]; then export PDB_LIB=$lib; else unset PDB_LIB; fi
  timeout 200 python3 bench.py $H "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $label: %.2f M env-steps/s, %.1f us per partition tick, contact-pass cars %s' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d.get('contact_pass_cars')))" || echo "  $label: FAILED"; }
for rep in 1 2; do
line libpdbatch_r2.so "r2 final, 4096 flat, 3 partitions"
line "" "now"
line libpdbatch_licm.so "now, machine LICM off"
done
line libpdbatch_r2.so "r2 final, 16384 flat" --cars 16384
line "" "now, 16384 flat" --cars 16384
line libpdbatch_licm.so "now LICM off, 16384 flat" --cars 16384
line libpdbatch_r2.so "r2 final, 4096 no body contacts" --no-body-contacts
line "" "now, 4096 no body contacts" --no-body-contacts
S="--steps 300 --warmup 50 --settle 200"
line libpdbatch_r2.so "r2 final, touge 16384 feedback" --workload touge --cars 16384 $S
line "" "now, touge 16384 feedback" --workload touge --cars 16384 $S
line libpdbatch_licm.so "now LICM off, touge 16384 feedback" --workload touge --cars 16384 $S
line libpdbatch_r2.so "r2 final, touge walls 16384 episodes" --workload touge --walls --episodes --cars 16384 $S
line "" "now, touge walls 16384 episodes" --workload touge --walls --episodes --cars 16384 $S
line libpdbatch_licm.so "now LICM off, touge walls 16384 episodes" --workload touge --walls --episodes --cars 16384 $S
line libpdbatch_r2.so "r2 final, touge walls 4096 feedback (reset-free)" --workload touge --walls --policy feedback --cars 4096 $S
line "" "now, touge walls 4096 feedback (reset-free)" --workload touge --walls --policy feedback --cars 4096 $S
line libpdbatch_licm.so "now LICM off, touge walls 4096 feedback (reset-free)" --workload touge --walls --policy feedback --cars 4096 $S
line "" "now, playground 16384 episodes" --workload playground --episodes --cars 16384 $S
line libpdbatch_licm.so "now LICM off, playground 16384 episodes" --workload playground --episodes --cars 16384 $S
line "" "now, playground 16384 mlp" --workload playground --policy mlp --cars 16384 $S
line libpdbatch_licm.so "now LICM off, playground 16384 mlp" --workload playground --policy mlp --cars 16384 $S
