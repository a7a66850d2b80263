#!/usr/bin/env bash
# Short GPU-box visit: smoke, GPU parity tests, default bench.  Logs -> gpurun_out/.
# Usage:  gpurun --timeout 1200 -- 'bash tools/gpu_quick.sh [tag] [extra pytest args]'
tag="${1:-r02}"; shift || true
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke_${tag}.log" 2>&1; echo "smoke rc=$?"; tail -3 "$out/smoke_${tag}.log"
echo "== pytest -m gpu"; rm -f "$out/parity_report.jsonl"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" > "$out/pytest_gpu_${tag}.log" 2>&1; echo "pytest rc=$?"; tail -40 "$out/pytest_gpu_${tag}.log"
cp "$out/parity_report.jsonl" "$out/parity_report_${tag}.jsonl" 2>/dev/null
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 > "$out/bench_${tag}.json" 2> "$out/bench_${tag}.err"; echo "bench rc=$?"; cut -c1-3000 "$out/bench_${tag}.json"; tail -3 "$out/bench_${tag}.err"
