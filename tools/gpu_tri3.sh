#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for cfg in panoptic shelf campus; do for b in 8; do
  echo -n "$cfg B=$b  lane-per-voxel: "; CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  echo -n "$cfg B=$b  quad packed:    "; FVP_TRIPLANE_QUAD=1 CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
done; done
echo -n "panoptic B=1 quad packed: "; FVP_TRIPLANE_QUAD=1 B=1 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane" 
for ab in 1 4 5; do echo -n "quad FVP_TRI_ABLATE=$ab  "; FVP_TRIPLANE_QUAD=1 FVP_TRI_ABLATE=$ab B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep project_triplane; done
