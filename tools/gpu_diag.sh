cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo "=== margins"; python tools/bench_conv.py --frames 8
echo "=== ablate 11"; FVP_CONV_ABLATE=11 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 3|op 4 |op 9 |op14|op19|total"
echo "=== bench"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-prof
) > gpurun_out/conv_diag15.log 2>&1
