#!/usr/bin/env bash
# One GPU-box visit: smoke, GPU parity tests, bench, rocprofv3 kernel trace.  Logs -> gpurun_out/.
# Usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/gpu_check.sh [tag]'
tag="${1:-r01}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > "$out/gpu.txt"
lscpu | grep -E "Model name|^CPU\(s\)|Socket|Core" >> "$out/gpu.txt"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$out/smoke.log"
echo "== pytest -m gpu"; rm -f "$out/parity_report.jsonl"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -25 "$out/pytest_gpu.log"
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 > "$out/bench_${tag}.json" 2> "$out/bench_${tag}.err"; echo "bench rc=$?"; cat "$out/bench_${tag}.json"; tail -3 "$out/bench_${tag}.err"
# sweeps: batch size at the default pipeline depth (2 batches in flight), and pipeline depth at B = 8 / B = 1
for spec in "1 3" "2 3" "16 3" "32 3" "8 1" "8 2" "1 1" "1 4"; do
  set -- $spec; b=$1; st=$2
  timeout 600 python bench.py --steps 10 --warmup 3 --batch $b --streams $st --no-cpu-baseline --no-extra > "$out/bench_${tag}_b${b}_s${st}.json" 2>> "$out/bench_${tag}.err"
  python - "$out/bench_${tag}_b${b}_s${st}.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("batch", d["config"]["frames_per_gpu_per_step"], "in flight", d["config"]["batches_in_flight"], "frames/s %.1f" % d["value"], "ms/step %.2f" % d["ms_per_step"], {k: round(v["ms_total"], 2) for k, v in d["kernels"].items()})
except Exception as e:
    print("bench sweep failed", e)
PY
done
echo "== backbone (next row f-1)"
timeout 600 python tools/bench_backbone.py --images 40 --iters 3 --per-op > "$out/backbone_per_op_${tag}.log" 2>&1; tail -2 "$out/backbone_per_op_${tag}.log"
timeout 600 python bench.py --backbone --steps 8 --warmup 2 --streams 2 --no-cpu-baseline --no-extra > "$out/bench_${tag}_e2e_b8_s2.json" 2>> "$out/bench_${tag}.err"
timeout 600 python bench.py --backbone --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-extra > "$out/bench_${tag}_e2e_b8_s1.json" 2>> "$out/bench_${tag}.err"
python - "$out/bench_${tag}_e2e_b8_s2.json" "$out/bench_${tag}_e2e_b8_s1.json" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f)); print("end-to-end", d["config"]["input"][:40], "in flight", d["config"]["batches_in_flight"], "frames/s %.1f" % d["value"])
    except Exception as e:
        print("e2e bench failed", e)
PY
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}" -o trace -- python "$root/bench.py" --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-prof --no-mpjpe --no-extra > "$out/rocprof_${tag}.log" 2>&1; echo "rocprof rc=$?"
find "$out/prof_${tag}" -name "*kernel_stats*.csv" | head -1 | xargs -r head -40 | cut -c1-220
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_bb" -o trace -- python "$root/tools/bench_backbone.py" --images 40 --iters 3 > "$out/rocprof_${tag}_bb.log" 2>&1; echo "rocprof backbone rc=$?"
