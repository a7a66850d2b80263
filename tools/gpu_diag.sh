cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "=== precomputed staging"; python tools/bench_conv.py --frames 8
echo "=== LDS 53"; FVP_CONV_LDS_KB=53 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 3|op 4 |op 9 |op14|op19|total"
echo "=== ablate 3"; FVP_CONV_ABLATE=3 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 3|op 4 |op 9 |op14|op19|total"
echo "=== ablate 8"; FVP_CONV_ABLATE=8 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 3|op 4 |op 9 |op14|op19|total"
echo "=== frames 16"; python tools/bench_conv.py --frames 16 | grep -E "op 0|op 3|op 4 |op 9 |op14|op19|total"
echo "=== bench"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline
) > gpurun_out/conv_diag7.log 2>&1
