#!/bin/bash
# Diagnostic (GPU box): the bench line's workload against ticks per launch and partitions.  Every command under its own timeout.
H="--no-cpu-baseline --no-extra"
line() { timeout 200 python3 bench.py $H ${@:2} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1: %.2f M env-steps/s, %.1f us per partition tick, repeats %d' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d['repeats']))" || echo "  $1: FAILED"; }
for t in 1 4 8 16; do line "4096 flat, 3 partitions, $t ticks/launch, 3000 steps" --ticks-per-launch $t --steps 3000 --warmup 333; done
for t in 1 8; do line "4096 flat, 3 partitions, $t ticks/launch, driver-style 20 steps" --ticks-per-launch $t --steps 20 --warmup 5; done
for p in 1 2; do line "4096 flat, $p partitions, 8 ticks/launch" --partitions $p --ticks-per-launch 8 --steps 3000 --warmup 333; done
line "4096 flat, no body contacts, 8 ticks/launch" --no-body-contacts --ticks-per-launch 8 --steps 3000 --warmup 333
line "4096 flat, no body contacts, 1 tick/launch" --no-body-contacts --ticks-per-launch 1 --steps 3000 --warmup 333
for t in 1 8; do line "16384 flat, $t ticks/launch" --cars 16384 --ticks-per-launch $t --steps 1500 --warmup 200; done
line "8192 gather k=1 + scatter (per-partition exchange)" --cars 8192 --steps 300 --warmup 50 --force-gather --gather-ticks 1 --scatter-actions
line "8192 gather k=32" --cars 8192 --steps 600 --warmup 100 --force-gather --gather-ticks 32
line "8192 no gather" --cars 8192 --steps 600 --warmup 100
