#!/usr/bin/env bash
# round 6, visit L: the five-quad form of the fused projection (JP = 20: Shelf / Campus) - GPU suite, then Shelf / Campus step and kernel time with and without it
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; cd "$root"; export TMPDIR=/tmp
rm -f "$out/parity_report.jsonl"
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$out/l.log" 2>&1; grep -E "passed|failed|^E " "$out/l.log" | head -5
for cfgn in shelf campus; do
  echo -n "$cfgn product: "
  timeout 300 python bench.py --config $cfgn --steps 60 --warmup 5 --no-extra --no-cpu-baseline --no-mpjpe 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s  tri-plane %.1f us/step' % (d['value'], 1e3*d['kernels']['per_step_ms']['project_triplane']))"
done
for sw in "X=1" "FVP_TRI_NO_Q5=1"; do for cfgn in shelf campus; do
  echo -n "$cfgn diag $sw: "
  env CFG=$cfgn B=8 FVP_LIB="$root/tests/diag/libfvp_hip_diag.so" $sw bash tools/kernel_class_times.sh 2>/dev/null | grep -i triplane | head -1
done; done
