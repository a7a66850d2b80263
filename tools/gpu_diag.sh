cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
echo "=== fuse s2"; python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-prof | cut -c80-130
echo "=== nofuse s2"; FVP_CONV_NO_POOL_FUSE=1 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-prof | cut -c80-130
done
) > gpurun_out/diag33.log 2>&1
