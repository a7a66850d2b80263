#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for ab in 0 1 2 3 4 8 12 15 7; do echo -n "FVP_C1D_ABLATE=$ab  "; FVP_C1D_ABLATE=$ab timeout 200 python tools/bench_c2c.py 10 2>&1 | grep -v amdgpu.ids; done
