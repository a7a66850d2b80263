#!/usr/bin/env bash
# GPU parity suite under the kernel-selection switches (every alternative form must pass the same tests)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
[[ -f tests/diag/libfvp_hip_diag.so ]] || tests/diag/build_diag.sh >/dev/null
# FVP_TEST_DIAG_LIB=1: tests/conftest.py points the package at the diagnostics build (the shipped library ignores FVP_* switches)
run() { echo -n "$* : "; env FVP_TEST_DIAG_LIB=1 "$@" timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -1; }
run X=1
run FVP_TRIPLANE_STAGED=1
run FVP_TRIPLANE_QUAD=1
run FVP_TRIPLANE_LANE=1 FVP_TRI_TWO_TILE=1
run FVP_TRIPLANE_QUAD=1 FVP_TRI_TWO_TILE=1 FVP_TRI_CAP_PX=200
run FVP_TRIPLANE_GATHER=1
run FVP_CONV_NO_KSPLIT=1 FVP_CONV_NO_HEAD_FUSE=1 FVP_CONV_NO_POOL_FUSE=1
run FVP_CONV_NO_WINO=1
run FVP_CONV_NO_REG=1
run FVP_CONV_REG_MIN_TILES=1
run FVP_WINO_HALF=1
run FVP_WINO_HALF=2
run FVP_WINO_QUARTER=2
run FVP_WINO_QUARTER=0
run FVP_WINO_W16=1 FVP_WINO_HALF=2
run FVP_TRI_ZRES=1
run FVP_CONV_NO_K7=1
run FVP_BB_NO_FUSE_STEM=1 FVP_BB_NO_FUSE_BLOCK=1
run FVP_TRI_NO_Q5=1
run FVP_WINO_GENERIC=1
