#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for spec in "panoptic128 1" "panoptic128 4" "shelf 8" "campus 8" "panoptic 8"; do set -- $spec
  echo "== $1 B=$2"; CFG=$1 B=$2 bash tools/kernel_class_times.sh 2>/dev/null | grep -v checksum
done
