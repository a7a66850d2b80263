#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
run() { echo "-- $*"; env "$@" timeout 200 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 --ops 0,2,8,14,20,23 2>&1 | grep -E "op " ; }
run X=1
run FVP_LIB=tools/scratch/libfvp_hip_occ2.so
run FVP_LIB=tools/scratch/libfvp_hip_occ4.so
run FVP_CONV_LDS_KB=32
run FVP_CONV_LDS_KB=128
run FVP_CONV_NO_PAIR=1
run FVP_CONV_NO_HEAD_FUSE=1
