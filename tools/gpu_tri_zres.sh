#!/usr/bin/env bash
# z-resident plane cells of the fused per-person projection (k_project_triplane_blk<.., ZRES>): parity, then the same-box
# A/B through the diagnostics build (FVP_TRI_ZRES=0 keeps the per-z-block cells)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
echo "== parity"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $out/zres_pytest.log 2>&1; tail -3 $out/zres_pytest.log | cut -c1-300
for cfg in panoptic shelf campus; do
  for z in 0 1; do
    echo -n "$cfg ZRES=$z: "; FVP_LIB=tests/diag/libfvp_hip_diag.so FVP_TRI_ZRES=$z CFG=$cfg B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|checksum" | tr '\n' ' '; echo
  done
done | tee $out/zres_ab.log
echo "== pipe"; timeout 200 python tools/bench_pipe.py --config panoptic --batch 8 --streams 4 --steps 100 2>&1 | tail -1 | tee $out/zres_pipe.log
