#!/usr/bin/env bash
# round 6, final visit at HEAD: what the driver runs (smoke, GPU suite, default bench) + the N > 1 code path forced on at world size 1
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
rm -f "$out/parity_report.jsonl"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py > "$out/bench_r06_final.json" 2> "$out/bench_r06_final.err"; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$out/bench_r06_final.json')); print('frames/s %.1f ms/step %.3f roofline %.3f traffic_src %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'][-70:])); print('long', d['config']['long_run']); print('e2e', d['e2e']['frames_per_s'], d['e2e']['backbone_ms_per_40_images'])"
FVP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline --no-mpjpe 2> /dev/null | tail -1 > "$out/bench_r06_forced_rccl_world1.json"; python -c "
import json; d=json.load(open('$out/bench_r06_forced_rccl_world1.json')); print('forced RCCL world 1: frames/s %.1f' % d['value'], d['config']['parallelism'][:60])"
