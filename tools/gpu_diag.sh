# Scratch script for one-off GPU-box experiments (gpurun -- 'bash tools/gpu_diag.sh'); logs go to gpurun_out/.
# The judged runs are tools/gpu_check.sh (smoke, GPU tests, bench sweeps, rocprofv3 traces) and tools/gpu_pmc.sh.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/diag.log 2>&1
