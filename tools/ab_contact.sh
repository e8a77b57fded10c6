#!/bin/bash
# Diagnostic (GPU box): the contact pass with three cars per workgroup (libpdbatch_cpb3.so) against the shipped one, kernel stats of two playground legs
for leg in "--episodes" "--policy mlp"; do
  for lib in "" "libpdbatch_cpb3.so"; do
    bash tools/kstats.sh "pg_$(echo $leg | tr -d ' -')_$(basename ${lib:-shipped} .so)" "$lib" --workload playground $leg --cars 16384 --steps 100 --warmup 20 --settle 200
  done
done
