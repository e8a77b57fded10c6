#!/bin/bash
# Diagnostic (GPU box): the contact pass's duration against its grid (PDB_CONTACT_GRID fixes the number of workgroups; they take the queued
# groups in turn) on the playground's reset-free leg -- does a workgroup's second group run faster than its first (code and tables warm)?
for g in 16 32 64 128 256 0; do
  if [ $g -gt 0 ]; then export PDB_CONTACT_GRID=$g; else unset PDB_CONTACT_GRID; fi
  bash tools/kstats.sh "grid_$g" "" --workload playground --policy mlp --cars 16384 --steps 100 --warmup 20 --settle 200 | grep "contact_kernel\|=="
done
