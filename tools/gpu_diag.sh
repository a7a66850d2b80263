cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
for w in 256 240 224 192 512; do
echo "=== wgs $w s3"; FVP_WINO_WGS=$w python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-prof | cut -c80-130
done
echo "=== wgs 512 s1";  FVP_WINO_WGS=512 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof| cut -c80-130
) > gpurun_out/diag38.log 2>&1
