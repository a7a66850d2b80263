#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "golden_case or fused_projection or batch_invariance or two_sequences or negative_bbox or empty or pipelined" 2>&1 | tail -3 | cut -c1-300
for cfg in panoptic shelf campus panoptic128; do b=8; [ $cfg = panoptic128 ] && b=1
  echo -n "$cfg B=$b lane-per-voxel:        "; CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  echo -n "$cfg B=$b quad:                  "; FVP_TRIPLANE_QUAD=1 CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  echo -n "$cfg B=$b quad, no 2-tile stage: "; FVP_TRI_ABLATE=16 FVP_TRIPLANE_QUAD=1 CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane" | tr '\n' ' '; echo
done
echo -n "panoptic B=1 lane: "; B=1 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
echo -n "panoptic B=1 quad: "; FVP_TRIPLANE_QUAD=1 B=1 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
