#!/bin/bash
# Diagnostic (GPU box): free-running partitions against the number of hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES, default 4).
H="--no-cpu-baseline --no-extra --steps 3000 --warmup 333"
line() { timeout 200 python3 bench.py $H "${@:2}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1: %.2f M env-steps/s, %.1f us per partition tick' % (d['value']/1e6, d['roofline']['kernel_avg_us']))" || echo "  $1: FAILED"; }
for q in 4 8; do for p in 3 4; do GPU_MAX_HW_QUEUES=$q line "GPU_MAX_HW_QUEUES=$q, $p partitions, 4096 cars" --partitions $p; done; done
GPU_MAX_HW_QUEUES=8 line "GPU_MAX_HW_QUEUES=8, 4 partitions, 16384 cars" --partitions 4 --cars 16384
GPU_MAX_HW_QUEUES=4 line "GPU_MAX_HW_QUEUES=4, 3 partitions, 16384 cars" --partitions 3 --cars 16384
line "driver-style: 3 partitions, 20 steps" --steps 20 --warmup 5
GPU_MAX_HW_QUEUES=8 line "driver-style: GPU_MAX_HW_QUEUES=8, 4 partitions, 20 steps" --partitions 4 --steps 20 --warmup 5
