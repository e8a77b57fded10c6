# usage: ab.sh variantA.so variantB.so ; "" = in-tree
run() { # lib label args...
  lib=$1; shift; label=$1; shift
  if [ -n "$lib" ]; then export PDB_LIB=$lib; else unset PDB_LIB; fi
  python bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$label', '$*', j['value'], j['ms_per_step'], j.get('roofline',{}).get('launch_us')); break
"
}
