leg() { label=$1; shift; python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-28s %7.2f M  %7.1f us/step  first pass %6.1f us' % ('$label', j['value']/1e6, j['ms_per_step']*1e3, j['roofline']['kernel_avg_us'])); break
"; }
for r in 1 2 3; do
  leg "law   headline_1500" --steps 1500 --warmup 200 --device-law
  leg "torch headline_1500" --steps 1500 --warmup 200 --no-device-law
  leg "law   headline_driver" --steps 20 --warmup 5 --device-law
  leg "torch headline_driver" --steps 20 --warmup 5 --no-device-law
done
