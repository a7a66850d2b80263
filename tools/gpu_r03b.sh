#!/usr/bin/env bash
# round-3 visit B: split-K direct conv (CenterNet small maps), sweep detail, B = 1 latency with / without hipGraph
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
echo "== pytest (split-K, sweep, golden, conv stacks)"; rm -f "$out/parity_report.jsonl"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "split_k or seed_sweep or golden_case or conv_stacks or batch_invariance or pipelined or hipgraph" > "$out/pytest_gpu_r03b.log" 2>&1; echo "pytest rc=$?"; tail -12 "$out/pytest_gpu_r03b.log" | cut -c1-600
cp "$out/parity_report.jsonl" "$out/parity_report_r03b.jsonl" 2>/dev/null
echo "== CenterNet per-op, B = 1 and 8, with and without split-K"
for ks in 0 1; do
  for b in 1 8; do
    FVP_CONV_NO_KSPLIT=$ks timeout 300 python tools/bench_conv.py --net center_net --frames $b --iters 10 2>&1 | grep -v amdgpu.ids > "$out/conv_center_b${b}_noks${ks}.log"; echo "no_ksplit=$ks B=$b: $(tail -1 $out/conv_center_b${b}_noks${ks}.log)"
  done
done
grep -E "k3x3 @(20|40)" "$out/conv_center_b1_noks0.log"
echo "== B = 1 latency: eager vs hipGraph; B = 8 serial"
timeout 300 python bench.py --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 eager  ms/step %.3f' % d['ms_per_step'])"
timeout 300 python bench.py --batch 1 --streams 1 --graph --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 graph  ms/step %.3f' % d['ms_per_step'])"
timeout 300 python bench.py --batch 8 --streams 1 --graph --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 graph serial ms/step %.3f  frames/s %.1f' % (d['ms_per_step'], d['value']))"
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-mpjpe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 default  ms/step %.3f  frames/s %.1f serial %.1f' % (d['ms_per_step'], d['value'], d['config']['frames_per_s_one_batch_at_a_time'])); print(d['kernels']['per_step_ms'])"
