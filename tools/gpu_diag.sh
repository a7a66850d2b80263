cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== conv per-op"; python tools/bench_conv.py --ops 0,1,2 2>&1 | grep " op"
echo "=== nopair"; FVP_CONV_NO_PAIR=1 python tools/bench_conv.py --ops 0 2>&1 | grep " op"
echo "=== center"; python tools/bench_conv.py --net center_net --ops 0,1,2,3 2>&1 | grep " op"
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "=== bench b8 noprof"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof | cut -c1-200
) > gpurun_out/diag21.log 2>&1
