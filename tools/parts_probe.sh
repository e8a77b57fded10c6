B="--cars 16384 --steps 100 --warmup 20 --settle 200 --no-cpu-baseline --no-extra --workload playground --policy mlp"
for p in 2 3 4; do
  timeout 200 python3 bench.py $B --partitions $p 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('partitions $p: %.2f M' % (d['value']/1e6))"
done
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python3 bench.py $B --partitions 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('partitions 4, GPU_MAX_HW_QUEUES=$q: %.2f M' % (d['value']/1e6))"
done
