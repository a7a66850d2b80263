#!/usr/bin/env bash
# k_conv7 (7x7 front conv on 16x16x4 tiles): parity, then same-box A/B against the pixel-pair form (FVP_CONV_NO_K7=1,
# diagnostics build), per op and pipelined
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
echo "== parity"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $out/k7_pytest.log 2>&1; tail -3 $out/k7_pytest.log | cut -c1-300
D=tests/diag/libfvp_hip_diag.so
for net in conv_net center_net; do
  for k in 0 1; do
    echo "-- $net NO_K7=$k"; FVP_LIB=$D FVP_CONV_NO_K7=$k timeout 200 python tools/bench_conv.py --net $net --frames 8 --iters 10 --ops 0 2>&1 | grep -E "op 0|total" | cut -c1-90
  done
done | tee $out/k7_ab.log
echo "-- B=1"; for k in 0 1; do FVP_LIB=$D FVP_CONV_NO_K7=$k timeout 200 python tools/bench_conv.py --net conv_net --frames 1 --iters 10 --ops 0 2>&1 | grep -E "op 0" | cut -c1-90; done | tee -a $out/k7_ab.log
echo "== pipe"
for k in 0 1; do FVP_LIB=$D FVP_CONV_NO_K7=$k timeout 200 python tools/bench_pipe.py --config panoptic --batch 8 --streams 4 --steps 100 2>&1 | tail -1; done | tee $out/k7_pipe.log
