# Scratch script for one-off GPU-box experiments (gpurun -- 'bash tools/gpu_diag.sh'); logs go to gpurun_out/.
# The judged runs are tools/gpu_check.sh (smoke, GPU tests, bench sweeps, rocprofv3 traces) and tools/gpu_pmc.sh.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json > gpurun_out/diag.log
