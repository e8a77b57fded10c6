# round 6 GPU job: occupancy curve, rocprofv3 kernel stats + PMC passes of the headline and of configs[1], the driver's own bench command; small summaries only come back
mkdir -p gpurun_out/r06_profiles
python3 tools/occupancy_curve.py > gpurun_out/r06_profiles/r06_occupancy_curve.txt 2>&1
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
bash tools/profile_round.sh r06_flat flat --workload flat > gpurun_out/r06_flat_profile_round.log 2>&1
cp profiles/r06_* gpurun_out/r06_profiles/ 2>/dev/null
for t in r06 r06_flat; do cp gpurun_out/$t/bench.json gpurun_out/r06_profiles/${t}_bench_3000.json 2>/dev/null; rm -rf gpurun_out/$t; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_profiles/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err
ls -la gpurun_out/r06_profiles
# the other configs' shapes (builder legs), on record
PDB_BENCH_EXTRA=configs4_playground_16384_mlp,episodes_4096,episodes_4096_reset_free,configs4_driftplayground_16384_mlp,configs2_16384,episodes_16384,episodes_16384_reset_free,configs4_nordring_16384_mlp,configs4_driftplayground_8192_host_mlp python bench.py --steps 300 --warmup 50 > gpurun_out/r06_profiles/r06_bench_line.json 2> gpurun_out/r06_bench_extra.err; cp gpurun_out/bench_extra.json gpurun_out/r06_profiles/r06_bench_extra.json 2>/dev/null
ls -la gpurun_out/r06_profiles | tail -5
