#!/usr/bin/env bash
# Phase stamps (s_memtime) of the Winograd K loop: needs tools/scratch/libfvp_hip_wtime.so (build_variant.sh wtime -DFVP_WINO_TIMING=1)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for ab in ${ABLATES:-0}; do
  echo "-- ablate $ab"
  FVP_CONV_ABLATE=$ab FVP_LIB=tools/scratch/libfvp_hip_wtime.so FVP_WINO_TIMING_PRINT=1 timeout 200 python tools/bench_conv.py --net conv_net --frames ${FRAMES:-8} --iters 1 2>&1 | grep -A2 -E "wino timing\]" | awk '/wino timing/{k=$0; c[k]++} c[k]==2' | sed 's/ |/\n      /g' | grep -v "^ *$" | cut -c1-200 | head -${LINES_MAX:-120}
done
