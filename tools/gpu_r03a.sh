#!/usr/bin/env bash
# round-3 visit A: parity suite + default bench line + per-op A/B of the Winograd epilogue (base = round-2 epilogue)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke_r03a.log" 2>&1; echo "smoke rc=$?"; tail -2 "$out/smoke_r03a.log"
echo "== pytest -m gpu"; rm -f "$out/parity_report.jsonl"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$out/pytest_gpu_r03a.log" 2>&1; echo "pytest rc=$?"; tail -30 "$out/pytest_gpu_r03a.log"
cp "$out/parity_report.jsonl" "$out/parity_report_r03a.jsonl" 2>/dev/null
echo "== per-op A/B"
for lib in "" tools/scratch/libfvp_hip_base.so; do
  echo "-- lib=${lib:-current}"
  FVP_LIB=$lib timeout 300 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 2>&1 | grep -v amdgpu.ids | tee "$out/conv_p2p_b8_${lib:+base}.log" | grep -E "k3x3|total"
  FVP_LIB=$lib B=8 timeout 300 bash tools/kernel_class_times.sh 2>&1 | grep -v amdgpu.ids
done
timeout 300 python tools/bench_conv.py --net center_net --frames 1 --iters 10 2>&1 | grep -v amdgpu.ids > "$out/conv_center_b1.log"; tail -1 "$out/conv_center_b1.log"
timeout 300 python tools/bench_conv.py --net conv_net --frames 1 --iters 10 2>&1 | grep -v amdgpu.ids > "$out/conv_p2p_b1.log"; tail -1 "$out/conv_p2p_b1.log"
timeout 300 python tools/bench_c2c.py 80 10 2>&1 | grep -v amdgpu.ids
echo "== bench (default line)"; timeout 1200 python bench.py > "$out/bench_r03a.json" 2> "$out/bench_r03a.err"; echo "bench rc=$?"; cut -c1-6000 "$out/bench_r03a.json"; tail -3 "$out/bench_r03a.err"
