#!/usr/bin/env bash
# fused-projection A/B through the per-class timers: current library vs FVP_LIB variants / env settings
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for cfg in panoptic shelf campus; do
  echo -n "$cfg default:        "; CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  echo -n "$cfg quad:           "; FVP_TRIPLANE_QUAD=1 CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane" | tr '\n' ' '; echo
  echo -n "$cfg quad two-tile:  "; FVP_TRIPLANE_QUAD=1 FVP_TRI_TWO_TILE=1 CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane" | tr '\n' ' '; echo
done
echo -n "panoptic quad cap 1 (all gathered): "; FVP_TRI_CAP_PX=1 B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
echo -n "panoptic B=1: "; B=1 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
