#!/usr/bin/env bash
# k_conv7 with 3 / 2 / 1 workgroups per CU (LDS request padded: FVP_K7_LDS_KB, diagnostics build) against the pixel-pair form: per op and pipelined
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out; D=tests/diag/libfvp_hip_diag.so
for e in "X=1" "FVP_K7_LDS_KB=56" "FVP_K7_LDS_KB=88" "FVP_CONV_NO_K7=1"; do
  echo "-- $e"; env FVP_LIB=$D $e timeout 200 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 --ops 0 2>&1 | grep -E "op 0" | cut -c1-90
  env FVP_LIB=$D $e timeout 200 python tools/bench_pipe.py --config panoptic --batch 8 --streams 4 --steps 150 2>&1 | tail -1
done | tee $out/k7d.log
