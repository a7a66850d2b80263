#!/usr/bin/env bash
# round 6, visit F: masked-tile Winograd shipped for CenterNet's 80- / 40-wide levels: full GPU suite, bench line, CenterNet per op
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
rm -f "$out/parity_report.jsonl"
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 3 > "$out/bench_r06f.json" 2> "$out/bench_r06f.err"; echo "bench rc=$?"
python - "$out/bench_r06f.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("frames/s %.1f ms/step %.3f long %s" % (d["value"], d["ms_per_step"], d["config"].get("long_run")))
print("per_step", d["kernels"]["per_step_ms"]); print("lat b1", d["latency_ms_b1_serial"]); print("pipe", {k: round(v["frames_per_s"], 1) for k, v in d["pipeline_launch_paths"].items()})
print("other", {k[:12]: round(v["frames_per_s"], 1) for k, v in d["other_configs"].items() if isinstance(v, dict)}); print("e2e", d["e2e"]["frames_per_s"])
PY
timeout 300 python tools/bench_conv.py --net center_net --frames 8 --iters 20 2>&1 | grep -v amdgpu > "$out/conv_per_op_centernet_b8_r06f.log"; tail -1 "$out/conv_per_op_centernet_b8_r06f.log"
timeout 300 python tools/bench_conv.py --net center_net --frames 1 --iters 20 2>&1 | grep -v amdgpu > "$out/conv_per_op_centernet_b1_r06f.log"; tail -1 "$out/conv_per_op_centernet_b1_r06f.log"
