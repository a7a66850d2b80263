#!/usr/bin/env bash
# round 6, visit A: GPU parity suite + default bench line + backbone per-op on the current tree
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
tag="${1:-r06a}"
rm -f "$out/parity_report.jsonl"
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$out/pytest_gpu_${tag}.log" 2>&1; echo "pytest rc=$?"; tail -5 "$out/pytest_gpu_${tag}.log"
timeout 600 python bench.py --steps 20 --warmup 3 > "$out/bench_${tag}.json" 2> "$out/bench_${tag}.err"; echo "bench rc=$?"
python - "$out/bench_${tag}.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("frames/s %.1f ms/step %.3f long %s" % (d["value"], d["ms_per_step"], d["config"].get("long_run")))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"]); print("per_step", d["kernels"]["per_step_ms"])
print("e2e", {k: d["e2e"].get(k) for k in ("frames_per_s", "backbone_ms_per_40_images", "backbone_tflops")}); print("lat b1", d["latency_ms_b1_serial"])
print("other", {k[:12]: round(v["frames_per_s"], 1) for k, v in d["other_configs"].items() if isinstance(v, dict)})
PY
timeout 600 python tools/bench_backbone.py --images 40 --iters 3 --per-op > "$out/backbone_per_op_${tag}.log" 2>&1; tail -3 "$out/backbone_per_op_${tag}.log"
