cd $GRAFT_REPO_ROOT
out=gpurun_out; tag=r01j
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout 600 python tools/bench_backbone.py --images 40 --iters 3 --per-op > "$out/backbone_per_op_${tag}.log" 2>&1; tail -2 "$out/backbone_per_op_${tag}.log"
timeout 600 python bench.py --backbone --steps 8 --warmup 2 --streams 2 --no-cpu-baseline > "$out/bench_${tag}_e2e_b8_s2.json" 2> /dev/null
timeout 600 python bench.py --backbone --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > "$out/bench_${tag}_e2e_b8_s1.json" 2> /dev/null
timeout 600 python bench.py --backbone --steps 8 --warmup 2 --streams 3 --no-cpu-baseline > "$out/bench_${tag}_e2e_b8_s3.json" 2> /dev/null
cut -c80-130 $out/bench_${tag}_e2e_b8_s1.json $out/bench_${tag}_e2e_b8_s2.json $out/bench_${tag}_e2e_b8_s3.json
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof_${tag}_bb" -o trace -- python "$GRAFT_REPO_ROOT/tools/bench_backbone.py" --images 40 --iters 3 > /dev/null 2>&1; echo "rocprof backbone rc=$?"
