cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== baseline"; python tools/bench_conv.py --frames 8
for ab in 1 2 3 4 8 12 7; do echo "=== ABLATE=$ab"; FVP_CONV_ABLATE=$ab python tools/bench_conv.py --frames 8 | grep -E "op 0|op 2 |op 4 |op 9 |op14|op19|op35|total"; done
echo "=== LDS 32KB"; FVP_CONV_LDS_KB=32 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 2 |op 4 |op 9 |op14|op19|total"
echo "=== LDS 48KB"; FVP_CONV_LDS_KB=48 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 2 |op 4 |op 9 |op14|op19|total"
echo "=== PB 2"; FVP_CONV_PB=2 python tools/bench_conv.py --frames 8 | grep -E "op 0|op 2 |op 4 |op 9 |op14|op19|total"
echo "=== c2c"; python tools/bench_conv.py --net c2c_net --frames 8 | tail -3
echo "=== center"; python tools/bench_conv.py --net center_net --frames 8 | tail -3
) > gpurun_out/conv_diag.log 2>&1
