#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; cd "$root"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
bash tools/gpu_switch_matrix.sh > "$out/switch_matrix_r06.log" 2>&1; cat "$out/switch_matrix_r06.log"
