cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
for i in 1 2; do
echo "=== quad s1"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof| cut -c80-130
echo "=== noquad s1"; FVP_WHOLE_NO_QUAD=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-prof| cut -c80-130
echo "=== quad s2"; python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-prof | cut -c80-130
echo "=== noquad s2"; FVP_WHOLE_NO_QUAD=1 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-prof | cut -c80-130
done
) > gpurun_out/diag31.log 2>&1
