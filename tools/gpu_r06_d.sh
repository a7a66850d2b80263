#!/usr/bin/env bash
# round 6, visit D: CenterNet through the masked-tile Winograd form (FVP_WINO_GENERIC, diagnostics build) against the shipped direct form
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for fr in 8 1; do for g in 0 1; do
  echo "== frames $fr FVP_WINO_GENERIC=$g"
  FVP_LIB="$root/tests/diag/libfvp_hip_diag.so" FVP_WINO_GENERIC=$g timeout 300 python tools/bench_conv.py --net center_net --frames $fr --iters 20 2>&1 | grep -E "k3x3|total"
done; done
