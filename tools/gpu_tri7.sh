#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for cfg in panoptic shelf; do
  echo -n "$cfg check-first: "; CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  echo -n "$cfg plain atomics: "; FVP_LIB=tools/scratch/libfvp_hip_nocheck.so CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
done
echo -n "panoptic B=1: "; B=1 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane"
