#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
for q in 5 8 12 24 48; do
  echo -n "GPU_MAX_HW_QUEUES=$q : "
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline --no-mpjpe --no-prof 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s' % d['value'])"
done
