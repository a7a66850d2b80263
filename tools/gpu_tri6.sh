#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "golden_case or fused_projection or batch_invariance or two_sequences or negative_bbox or empty or pipelined or precomputed" 2>&1 | tail -3 | cut -c1-300
for cfg in panoptic shelf campus panoptic128; do b=8; [ $cfg = panoptic128 ] && b=1
  echo -n "$cfg B=$b default:   "; CFG=$cfg B=$b timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
done
echo -n "panoptic B=8 forced lane form: "; FVP_TRIPLANE_LANE=1 B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
echo -n "shelf B=8 quad form:           "; FVP_TRIPLANE_QUAD=1 CFG=shelf B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
echo -n "panoptic B=1: "; B=1 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
for ab in 4 5; do echo -n "FVP_TRI_ABLATE=$ab  "; FVP_TRI_ABLATE=$ab B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep project_triplane; done
