#!/usr/bin/env bash
# one-off knob sweep through tools/bench_conv.py:  gpurun -- 'bash tools/gpu_knobs.sh "<ops>" "ENV=val" "ENV=val ENV2=val" ...'
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
ops="$1"; shift
run() { echo "-- $*"; env "$@" timeout 200 python tools/bench_conv.py --net ${NET:-conv_net} --frames ${FRAMES:-8} --iters 10 --ops "$ops" ${SPAN:+--span $SPAN} 2>&1 | grep -E "^ +(op|span)" ; }
run X=1
for v in "$@"; do run $v; done
