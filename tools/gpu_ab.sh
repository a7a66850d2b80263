#!/usr/bin/env bash
# Same-box A/B of library variants: per-op timings of a conv stack (twice, interleaved) and the pipelined bench line.
#   gpurun -- 'bash tools/gpu_ab.sh "<pytest -k expr or empty>" "" tools/scratch/libfvp_hip_X.so ...'   ("" = the product library)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
k="$1"; shift
[[ -n "$k" ]] && timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "$k" 2>&1 | tail -2
for i in 1 2; do for v in "$@"; do echo "-- ${v:-product}"; FVP_LIB=$v timeout 200 python tools/bench_conv.py --net ${NET:-conv_net} --frames ${FRAMES:-8} --iters 10 2>&1 | grep -E "k3x3|total" | cut -c1-60; done; done
if [[ -n "${BENCH:-}" ]]; then
  for i in 1 2; do for v in "$@"; do FVP_LIB=$v timeout 300 python tools/bench_pipe.py 2>&1 | tail -1; done; done
fi
