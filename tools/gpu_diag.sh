cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "=== bench b8"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline
echo "=== bench b8 noprof"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof | cut -c1-200
echo "=== bench b32 noprof"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof --batch 32 | cut -c1-200
) > gpurun_out/diag16.log 2>&1
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fused -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
grep -E "fused|project|softarg|nms" $GRAFT_REPO_ROOT/gpurun_out/prof_fused/trace_kernel_stats.csv | cut -c1-160 >> $GRAFT_REPO_ROOT/gpurun_out/diag16.log
