#!/bin/bash
# Run on the GPU box (through gpurun; every command under its own timeout): bench line, rocprofv3 kernel stats, and the PMC passes that DESIGN.md /
# bench.py's roofline block cite.  Usage: bash tools/profile_round.sh <tag> [workload key] [bench args of that workload...]   (outputs under gpurun_out/<tag>/)
#   bash tools/profile_round.sh r05                          the headline (BASELINE configs[2] as worded)
#   bash tools/profile_round.sh r05_flat flat --workload flat    configs[1]
# Counter passes carry --pmc only (never with a trace domain), the program itself follows `--`.
set -u
TAG=${1:-r05}
KEY=${2:-ek_akina}
shift; shift
ARGS="$* --no-secondary --no-cpu-baseline"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${PDB_PROFILE_SKIP_BENCH:-0}" != "1" ]; then
  timeout 1500 python3 bench.py --steps 3000 --warmup 333 $* > $OUT/bench.json 2> $OUT/bench.err
  tail -1 $OUT/bench.json | cut -c1-300
fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --steps 300 --warmup 50 $ARGS > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 bench.py --steps 100 --warmup 20 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 bench.py --steps 100 --warmup 20 $ARGS > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o run -- python3 bench.py --steps 100 --warmup 20 $ARGS > $OUT/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o run -- python3 bench.py --steps 100 --warmup 20 $ARGS > $OUT/pmc_sq2.log 2>&1
# round 6: where the issue-stalled share comes from (24 % on ek_akina against 7 % on the plane): waits for LDS results and vector-memory reads
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq3 -o run -- python3 bench.py --steps 100 --warmup 20 $ARGS > $OUT/pmc_sq3.log 2>&1
# issue model: measured cycles per wave-instruction (tools/issue_rate.hip), plain and under the SQ counters
if [ -x tools/issue_rate ] && [ "${PDB_PROFILE_ISSUE:-0}" = "1" ]; then
  timeout 300 tools/issue_rate > $OUT/issue_rate.txt 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_issue -o run -- tools/issue_rate > $OUT/pmc_issue.log 2>&1
fi
python3 tools/summarise_profiles.py $TAG $KEY
