#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for cfg in panoptic shelf; do
for ab in 0 8; do
  echo -n "$cfg ablate=$ab lane-per-voxel: "; FVP_TRI_ABLATE=$ab CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
  echo -n "$cfg ablate=$ab quad:           "; FVP_TRIPLANE_QUAD=1 FVP_TRI_ABLATE=$ab CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
done; done
