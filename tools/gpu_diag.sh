cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6
for cfgx in "2 1" "4 2" "8 2"; do set -- $cfgx
echo "=== e2e b$1 s$2"; timeout 300 python bench.py --backbone --steps 6 --warmup 2 --no-cpu-baseline --streams $2 --batch $1 2>&1 | tail -1 | cut -c80-140
done
timeout 300 python tools/bench_backbone.py --images 40 --iters 3 2>&1 | tail -1
) > gpurun_out/diag49.log 2>&1
