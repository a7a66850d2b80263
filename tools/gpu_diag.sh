cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== conv per-op"; python tools/bench_conv.py 2>&1 | tail -28
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "=== bench b8 noprof"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof | cut -c1-200
echo "=== bench b1 noprof"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-prof --batch 1 | cut -c1-200
echo "=== bench b32 noprof"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --batch 32 | cut -c1-200
) > gpurun_out/diag19.log 2>&1
