#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for w in 256 248 240 224; do for st in 3 4; do
echo -n "FVP_WINO_WGS=$w streams=$st  "; FVP_WINO_WGS=$w timeout 200 python bench.py --steps 60 --warmup 5 --streams $st --no-cpu-baseline --no-extra --no-mpjpe --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.3f  frames/s %.1f' % (d['ms_per_step'], d['value']))"
done; done
