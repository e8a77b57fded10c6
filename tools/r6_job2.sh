# round 6 final GPU job (after the device law and the exact-row classes): rocprofv3 kernel stats + PMC passes of the headline and of configs[1], the driver's own bench
# command, the builder legs, the MLP stand-in with its ReLUs in the GEMM epilogue; small summaries only come back
mkdir -p gpurun_out/r06_profiles
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
bash tools/profile_round.sh r06_flat flat --workload flat > gpurun_out/r06_flat_profile_round.log 2>&1
cp profiles/r06_* gpurun_out/r06_profiles/ 2>/dev/null
for t in r06 r06_flat; do cp gpurun_out/$t/bench.json gpurun_out/r06_profiles/${t}_bench_3000.json 2>/dev/null; rm -rf gpurun_out/$t; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_profiles/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err
PDB_BENCH_EXTRA=configs4_playground_16384_mlp,episodes_4096,episodes_4096_reset_free,configs4_driftplayground_16384_mlp,configs2_16384,episodes_16384,episodes_16384_reset_free,configs4_nordring_16384_mlp,configs4_nordring_16384_feedback,configs4_driftplayground_8192_host_mlp python bench.py --steps 300 --warmup 50 > gpurun_out/r06_profiles/r06_bench_line.json 2> gpurun_out/r06_bench_extra.err; cp gpurun_out/bench_extra.json gpurun_out/r06_profiles/r06_bench_extra.json 2>/dev/null
leg() { label=$1; shift; python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('%-44s %7.2f M  %7.1f us/step' % ('$label', j['value']/1e6, j['ms_per_step']*1e3)); break
"; }
{ for r in 1 2; do
  leg "playground 16384 mlp, six launches" --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200
  leg "playground 16384 mlp, ReLU in the epilogue" --workload playground --cars 16384 --policy mlp --steps 300 --warmup 50 --settle 200 --mlp-fused-relu
done; leg "flat 16384" --workload flat --cars 16384 --steps 1500 --warmup 200; leg "flat 16384" --workload flat --cars 16384 --steps 1500 --warmup 200; } > gpurun_out/r06_profiles/r06_mlp_epilogue_ab.txt 2>&1
ls -la gpurun_out/r06_profiles | tail -20
