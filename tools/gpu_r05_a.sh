#!/usr/bin/env bash
# round-5 visit A: Winograd phase stamps, per-op baselines, pipelined submit/drain at B = 8 / B = 1 / Campus
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
echo "== wino timing"; FRAMES=8 bash tools/gpu_wino_timing.sh > $out/r05a_wino_timing.log 2>&1; tail -60 $out/r05a_wino_timing.log
echo "== per-op p2pnet"; timeout 200 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 > $out/r05a_p2p.log 2>&1; cat $out/r05a_p2p.log | cut -c1-110
echo "== per-op centernet"; timeout 200 python tools/bench_conv.py --net center_net --frames 8 --iters 10 > $out/r05a_cn.log 2>&1; tail -32 $out/r05a_cn.log | cut -c1-110
echo "== pipe"
for cfg in "panoptic 8" "panoptic 1" "campus 8"; do set -- $cfg; timeout 200 python tools/bench_pipe.py --config $1 --batch $2 --streams 4 --steps 100 2>&1 | tail -2; done | tee $out/r05a_pipe.log
