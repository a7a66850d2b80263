#!/usr/bin/env bash
# round-3 visit C: fused C2CNet rewrite + split-K restricted to the 20x20 level
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
echo "== pytest -m gpu (all)"; rm -f "$out/parity_report.jsonl"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$out/pytest_gpu_r03c.log" 2>&1; echo "pytest rc=$?"; tail -6 "$out/pytest_gpu_r03c.log" | cut -c1-400
cp "$out/parity_report.jsonl" "$out/parity_report_r03c.jsonl" 2>/dev/null
echo "== c2c"; timeout 300 python tools/bench_c2c.py 80 10 2>&1 | grep -v amdgpu.ids
FVP_LIB=tools/scratch/libfvp_hip_base.so timeout 300 python tools/bench_c2c.py 80 10 2>&1 | grep -v amdgpu.ids
for b in 1 8; do timeout 300 python tools/bench_conv.py --net center_net --frames $b --iters 10 2>&1 | grep -v amdgpu.ids > "$out/conv_center_b${b}_r03c.log"; echo "CenterNet B=$b: $(tail -1 $out/conv_center_b${b}_r03c.log)"; done
echo "== bench"
timeout 300 python bench.py --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 serial ms/step %.3f' % d['ms_per_step'], d['kernels']['per_step_ms'])"
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 default  ms/step %.3f  frames/s %.1f serial %.1f' % (d['ms_per_step'], d['value'], d['config']['frames_per_s_one_batch_at_a_time'])); print(d['kernels']['per_step_ms'])"
