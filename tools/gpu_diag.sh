cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "backbone or images" 2>&1 | tail -2
timeout 300 python tools/bench_backbone.py --images 40 --iters 3 --per-op 2>&1 | grep -E "op5[4-7] |images|total"
echo "=== e2e b8 s2"; timeout 300 python bench.py --backbone --steps 8 --warmup 2 --no-cpu-baseline --streams 2 | cut -c80-130
) > gpurun_out/diag56.log 2>&1
