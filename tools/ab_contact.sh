#!/bin/bash
# Diagnostic (GPU box): kernel stats of the playground legs (optionally against a library variant: ab_contact.sh libpdbatch_x.so)
for leg in "--episodes" "--policy mlp"; do
  for lib in "" $1; do
    bash tools/kstats.sh "pg_$(echo $leg | tr -d ' -')_$(basename ${lib:-shipped} .so)" "$lib" --workload playground $leg --cars 16384 --steps 100 --warmup 20 --settle 200
  done
done
