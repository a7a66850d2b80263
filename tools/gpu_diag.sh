cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== conv per-op"; python tools/bench_conv.py --ops 20,23 2>&1 | grep " op"
echo "=== nopair"; FVP_CONV_NO_PAIR=1 python tools/bench_conv.py --ops 20,23 2>&1 | grep " op"
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "=== bench b8"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline | cut -c1-1200
echo "=== bench b8 noprof"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof | cut -c1-200
) > gpurun_out/diag23.log 2>&1
