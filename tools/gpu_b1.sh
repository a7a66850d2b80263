#!/usr/bin/env bash
# B = 1 latency A/B: per-op P2PNet / CenterNet at one frame and the serial B = 1 bench
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "golden_case or batch_invariance or conv_stacks or pipelined or hipgraph" 2>&1 | tail -2 | cut -c1-200
for e in "X=1" "FVP_WINO_NO_DEEP_RING=1"; do
  echo "-- $e"; env $e timeout 200 python tools/bench_conv.py --net conv_net --frames 1 --iters 10 2>&1 | grep -E "k3x3|total" | cut -c1-70
  env $e timeout 300 python bench.py --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 serial ms/step %.3f' % d['ms_per_step'], d['kernels']['per_step_ms'])"
done
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 default  ms/step %.3f  frames/s %.1f' % (d['ms_per_step'], d['value']))"
timeout 300 python bench.py --batch 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 3 in flight  ms/step %.3f  frames/s %.1f' % (d['ms_per_step'], d['value']))"
