#!/usr/bin/env bash
# round 6, visit H: the per-batch result gather at world size 1 (RCCL path forced on): host-issued (default) against the GPU-side event wait
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "pipelin or Pipelin or graph or concurrency or images" 2>&1 | tail -3
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline --no-mpjpe --no-prof 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host']; print('%.1f frames/s  host submit %.3f ms/step  back-pressure wait %.3f ms/step  drain %.1f ms' % (d['value'], h['submit_ms_per_step'], h['backpressure_wait_ms_per_step'], h['drain_ms']))"; }
run X=1
run FVP_BENCH_FORCE_DIST=1
run FVP_BENCH_FORCE_DIST=1 FVP_GATHER_ISSUE=stream
run X=1
run FVP_BENCH_FORCE_DIST=1
run FVP_BENCH_FORCE_DIST=1 FVP_GATHER_ISSUE=stream
