cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6
echo "=== backbone"; timeout 600 python tools/bench_backbone.py --images 5 2>&1 | tail -2
timeout 600 python tools/bench_backbone.py --images 40 --iters 3 2>&1 | tail -1
) > gpurun_out/diag43.log 2>&1
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bb -o trace -- python $GRAFT_REPO_ROOT/tools/bench_backbone.py --images 10 --iters 3 > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_bb/trace_kernel_stats.csv | cut -c1-160 >> $GRAFT_REPO_ROOT/gpurun_out/diag43.log
