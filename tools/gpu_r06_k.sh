#!/usr/bin/env bash
# round 6, visit K: register-direct vs LDS-DMA form for the small 1x1 / transposed launches (threshold re-check)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for net in center_net conv_net; do for fr in 8 1; do for sw in "X=1" "FVP_CONV_REG_MIN_TILES=1" "FVP_CONV_NO_REG=1"; do
  echo "== $net frames $fr $sw"
  env FVP_LIB="$root/tests/diag/libfvp_hip_diag.so" $sw timeout 300 python tools/bench_conv.py --net $net --frames $fr --iters 20 2>&1 | grep -E "k1x1|convT|total"
done; done; done
