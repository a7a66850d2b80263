#!/usr/bin/env bash
# round 6, visit G: unit-size thresholds of the Winograd kernel on CenterNet's sub-chip launches; pipeline depth after the CenterNet change
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for fr in 8 1; do for sw in "X=1" "FVP_WINO_HALF=1" "FVP_WINO_HALF=2" "FVP_WINO_QUARTER=2" "FVP_WINO_QUARTER=0"; do
  echo -n "frames $fr $sw : "
  env FVP_LIB="$root/tests/diag/libfvp_hip_diag.so" $sw timeout 300 python tools/bench_conv.py --net center_net --frames $fr --iters 20 2>&1 | grep -E "total"
done; done
for st in 3 4 5 6; do
  echo -n "streams $st : "
  timeout 300 python bench.py --steps 60 --warmup 5 --streams $st --no-cpu-baseline --no-extra --no-prof --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f frames/s' % d['value'])"
done
