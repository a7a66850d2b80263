cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
timeout 300 python tools/bench_backbone.py --images 40 --iters 3 2>&1 | tail -1
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "backbone or images" 2>&1 | tail -2
) > gpurun_out/diag51.log 2>&1
