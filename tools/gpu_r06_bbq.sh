#!/usr/bin/env bash
# quick backbone visit: fused-kernel parity test + per-op (fused groups) + pass time
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "backbone_fused or backbone_bf16" 2>&1 | tail -3
timeout 300 python tools/bench_backbone.py --images 40 --iters 5 --per-op 2>&1 | grep -E "fused group|ms/pass|total"
