#!/bin/bash
# Diagnostic (GPU box): the bench line's workload against partitions, region length and the collider switch; the 8192-car shard's collectives.  Every command under its own timeout.
H="--no-cpu-baseline --no-extra"
line() { timeout 200 python3 bench.py $H ${@:2} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1: %.2f M env-steps/s, %.1f us per partition tick, repeats %d' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d['repeats']))" || echo "  $1: FAILED"; }
line "4096 flat, 3 partitions, 3000 steps" --steps 3000 --warmup 333
line "4096 flat, 3 partitions, driver-style 20 steps" --steps 20 --warmup 5
for p in 1 2; do line "4096 flat, $p partitions" --partitions $p --steps 3000 --warmup 333; done
line "4096 flat, no body contacts" --no-body-contacts --steps 3000 --warmup 333
line "16384 flat" --cars 16384 --steps 1500 --warmup 200
line "8192 gather k=1 + scatter (per-partition exchange)" --cars 8192 --steps 300 --warmup 50 --force-gather --gather-ticks 1 --scatter-actions
line "8192 gather k=32" --cars 8192 --steps 600 --warmup 100 --force-gather --gather-ticks 32
line "8192 no gather" --cars 8192 --steps 600 --warmup 100
