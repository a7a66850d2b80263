cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
echo "=== persistent s1"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof| cut -c80-130
echo "=== one-unit s1"; FVP_WINO_WGS=1000000 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof| cut -c80-130
echo "=== persistent s3"; python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
echo "=== one-unit s3"; FVP_WINO_WGS=1000000 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
done
) > gpurun_out/diag37.log 2>&1
