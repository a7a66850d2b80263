#!/usr/bin/env bash
# round 6, visit C: Shelf R1 assertion under the default build and under FVP_CONV_NO_K7; backbone per-op after the epilogue change
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "seed_sweep or backbone" 2>&1 | tail -3
FVP_TEST_DIAG_LIB=1 FVP_CONV_NO_K7=1 timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "seed_sweep" 2>&1 | tail -3
timeout 300 python tools/bench_backbone.py --images 40 --iters 5 --per-op > "$out/backbone_per_op_r06c.log" 2>&1; cat "$out/backbone_per_op_r06c.log" | grep -v amdgpu
