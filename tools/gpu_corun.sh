#!/usr/bin/env bash
# Can the fused projection run in the 64 registers per lane a resident Winograd kernel leaves free?  Variant tri64 =
# -DFVP_TRI_BLK_OCC=8 (k_project_triplane_blk compiled for <= 64 registers).  Alone (per-class timers) and pipelined.
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
for lib in tests/diag/libfvp_hip_diag.so ${VARIANTS:-tools/scratch/libfvp_hip_tri64.so}; do
  echo "-- $lib"
  FVP_LIB=$lib CFG=panoptic B=8 timeout 200 bash tools/kernel_class_times.sh 2>&1 | grep -E "project_triplane|softargmax|checksum" | tr '\n' ' '; echo
  for e in "X=1" ${ENVS:-"FVP_WINO_LDS_KB=144"}; do
    echo -n "$e: "; env FVP_LIB=$lib $e timeout 200 python tools/bench_pipe.py --config panoptic --batch 8 --streams 4 --steps 150 2>&1 | tail -1
  done
done | tee $out/corun.log
