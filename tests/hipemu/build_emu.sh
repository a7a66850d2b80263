#!/usr/bin/env bash
# Build the CPU emulation of the HIP kernels (TEST INFRASTRUCTURE ONLY): the unmodified
# .hip sources are compiled for the host against tests/hipemu/hip/hip_runtime.h.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
csrc="${here}/../../faster-voxelpose_amd/csrc"
CXX="${HIPEMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
# -DFVP_DIAG=1: the emulated library is test infrastructure, its tests flip the kernel-selection switches through the environment
flags=(-x c++ -std=c++17 -O2 -fPIC -ffp-contract=off -DFVP_DIAG=1 -I"${here}" -Wno-unused-function -Wno-unknown-attributes)
objs=()
for s in fvp_capi fvp_project fvp_conv fvp_conv_wino fvp_conv1d_fused fvp_proposal fvp_joint fvp_heatmap fvp_backbone; do
  o="${here}/${s}.emu.o"
  if [[ ! -f "$o" || "$o" -ot "${csrc}/$s.hip" || "$o" -ot "${csrc}/fvp_common.h" || "$o" -ot "${csrc}/fvp_geom.h" \
        || "$o" -ot "${here}/hip/hip_runtime.h" || "$o" -ot "${here}/../../include/fvp.h" || "$o" -ot "${csrc}/fvp_asm.h" || "$o" -ot "${csrc}/fvp_conv_args.h" || "$o" -ot "${csrc}/fvp_conv_reg.h" || "$o" -ot "${csrc}/fvp_project_lds.h" || "$o" -ot "${csrc}/fvp_conv7.h" || "$o" -ot "${csrc}/fvp_backbone_fused.h" ]]; then
    "$CXX" "${flags[@]}" -c "${csrc}/$s.hip" -o "$o" &
  fi
  objs+=("$o")
done
o="${here}/hipemu.emu.o"
"$CXX" "${flags[@]}" -c "${here}/hipemu.cpp" -o "$o" &
objs+=("$o")
wait
"$CXX" -shared -fPIC "${objs[@]}" -o "${here}/libfvp_emu.so" -lpthread
echo "built ${here}/libfvp_emu.so"
