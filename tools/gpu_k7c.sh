#!/usr/bin/env bash
# k_conv7 phase ablations (variant builds, wrong results): 1 no tile DMA, 2 no LDS reads, 4 no weight loads
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
for v in "" ${VARIANTS:-k7a2 k7a6 k7a7}; do
  lib=tests/diag/libfvp_hip_diag.so; [ -n "$v" ] && lib=tools/scratch/libfvp_hip_$v.so
  echo -n "${v:-full}: "; FVP_LIB=$lib timeout 200 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 --ops 0 2>&1 | grep -E "op 0" | cut -c1-90
done | tee $out/k7c_ablate.log
