#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for ab in 0 1 2 3 4 5 7; do echo -n "FVP_TRI_ABLATE=$ab  "; FVP_TRI_ABLATE=$ab B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep project_triplane; done
echo -n "no fine cache  "; FVP_NO_FINE_CACHE=1 B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep project_triplane
