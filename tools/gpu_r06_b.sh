#!/usr/bin/env bash
# round 6, visit B: backbone only - fused-kernel parity tests, per-op times, e2e bench
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
tag="${1:-r06b}"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "backbone or images" > "$out/pytest_gpu_${tag}.log" 2>&1; echo "pytest rc=$?"; tail -8 "$out/pytest_gpu_${tag}.log"
timeout 600 python tools/bench_backbone.py --images 40 --iters 5 --per-op > "$out/backbone_per_op_${tag}.log" 2>&1; head -14 "$out/backbone_per_op_${tag}.log"; tail -3 "$out/backbone_per_op_${tag}.log"
timeout 600 python bench.py --backbone --steps 8 --warmup 2 --streams 2 --no-cpu-baseline --no-extra > "$out/bench_${tag}_e2e_b8_s2.json" 2> "$out/bench_${tag}.err"
python - "$out/bench_${tag}_e2e_b8_s2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("end-to-end frames/s %.1f" % d["value"])
PY
