cd $GRAFT_REPO_ROOT
out=gpurun_out; tag=r01k
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout 600 python tools/bench_backbone.py --images 40 --iters 3 --per-op > "$out/backbone_per_op_${tag}.log" 2>&1; tail -2 "$out/backbone_per_op_${tag}.log"
for st in 1 2 3; do
timeout 600 python bench.py --backbone --steps 10 --warmup 3 --streams $st --no-cpu-baseline > "$out/bench_${tag}_e2e_b8_s$st.json" 2> /dev/null
done
cut -c80-130 $out/bench_${tag}_e2e_b8_s1.json $out/bench_${tag}_e2e_b8_s2.json $out/bench_${tag}_e2e_b8_s3.json
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof_${tag}_bb" -o trace -- python "$GRAFT_REPO_ROOT/tools/bench_backbone.py" --images 40 --iters 3 > /dev/null 2>&1; echo "rocprof backbone rc=$?"
