#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "golden_case or fused_projection or batch_invariance or two_sequences or negative_bbox or empty or pipelined or seed_sweep" 2>&1 | tail -4 | cut -c1-300
for cfg in panoptic shelf campus; do for b in 8 1; do
  echo -n "$cfg B=$b  lane-per-voxel: "; CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  echo -n "$cfg B=$b  quad form:      "; FVP_TRIPLANE_QUAD=1 CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
done; done
for ab in 1 4 5 7; do echo -n "FVP_TRI_ABLATE=$ab  "; FVP_TRI_ABLATE=$ab B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep project_triplane; done
