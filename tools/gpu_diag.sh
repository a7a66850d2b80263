cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== conv per-op"; python tools/bench_conv.py 2>&1 | tail -45
echo "=== pytest conv"; timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo "=== bench b8 noprof"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof | cut -c1-200
) > gpurun_out/diag17.log 2>&1
