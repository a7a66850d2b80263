#!/usr/bin/env bash
# Winograd kernel A/B: parity tests with the current library, then per-op timings of the current library and of variants
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "golden_case or conv_stacks or batch_invariance or seed_sweep or pipelined or hipgraph or end_to_end" 2>&1 | tail -3 | cut -c1-300
run() { echo "-- ${1:-current}"; FVP_LIB=$1 timeout 200 python tools/bench_conv.py --net conv_net --frames ${FRAMES:-8} --iters 10 2>&1 | grep -E "k3x3|total" | cut -c1-60; }
run ""
for v in "$@"; do run tools/scratch/libfvp_hip_$v.so; done
