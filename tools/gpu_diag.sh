cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== conv per-op"; python tools/bench_conv.py --ops 3,4,5,9,10,15,16 2>&1 | grep " op"
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
echo "=== s1"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof| cut -c80-130
echo "=== s3"; python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
done
) > gpurun_out/diag39.log 2>&1
