cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest backbone"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k backbone 2>&1 | tail -3
timeout 600 python tools/bench_backbone.py --images 40 --iters 2 --per-op 2>&1 | tail -62) > gpurun_out/diag44.log 2>&1
