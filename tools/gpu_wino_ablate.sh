#!/usr/bin/env bash
# Winograd kernel anatomy: per-op timings of P2PNet's 3x3 layers with parts of the kernel switched off
# (FVP_CONV_ABLATE bits: 1 no DMA, 4 no MFMA, 8 no epilogue, ...).  Needs the ablation variant of the library:
#   tools/build_variant.sh wabl -DFVP_WINO_ABLATE=1      (round 5: the switches are compiled out of every other build)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for ab in 0 1 4 8 5 13; do
  echo "-- ablate $ab"
  FVP_LIB=tools/scratch/libfvp_hip_wabl.so FVP_CONV_ABLATE=$ab timeout 200 python tools/bench_conv.py --net conv_net --frames ${FRAMES:-8} --iters 10 2>&1 | grep -E "op ?(3|4|9|10|15|16) |total" | cut -c1-60
done
