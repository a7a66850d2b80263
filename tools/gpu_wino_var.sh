#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
run() { echo "-- $*"; env "$@" timeout 200 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 --ops 1,3,4,9,10,15,16 2>&1 | grep -E "op ?[0-9]" ; }
run X=1
for v in "$@"; do run FVP_LIB=tools/scratch/libfvp_hip_$v.so; done
