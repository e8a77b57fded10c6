#!/bin/bash
# Diagnostic (GPU box): rocprofv3 kernel stats of a bench command for a library variant.  usage: kstats.sh <tag> <PDB_LIB or ""> <bench args...>
TAG=$1; LIB=$2; shift 2
OUT=gpurun_out/kstats/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$LIB" ]; then export PDB_LIB=$LIB; else unset PDB_LIB; fi
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --no-cpu-baseline --no-extra "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
fs = glob.glob('$OUT/**/run_kernel_stats.csv', recursive=True) + glob.glob('$OUT/run_kernel_stats.csv')
if fs:
    rows=list(csv.DictReader(open(fs[0])))
    print('== $TAG')
    for r in rows[:3]: print('  %-44s calls %6s avg %9.1f us min %9.1f max %9.1f  total %5.1f%%' % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['Percentage'])))
else: print('== $TAG: no stats')
PY
