#!/usr/bin/env bash
# Build the HIP library for gfx950 (cross-compiles without a GPU).  Output: ../libfvp_hip.so
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libfvp_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
srcs=(fvp_capi.hip fvp_project.hip fvp_conv.hip fvp_conv1d_fused.hip fvp_proposal.hip fvp_joint.hip fvp_heatmap.hip fvp_backbone.hip)
objs=()
for s in "${srcs[@]}"; do
  o="${here}/${s%.hip}.o"
  if [[ ! -f "$o" || "$o" -ot "${here}/$s" || "$o" -ot "${here}/fvp_common.h" || "$o" -ot "${here}/fvp_geom.h" \
        || "$o" -ot "${here}/../../include/fvp.h" || "$o" -ot "${here}/fvp_conv_wino.h" || "$o" -ot "${here}/fvp_project_lds.h" ]]; then
    # geometry / proposal / fusion code mirrors the reference's separately-rounded fp32 ops:
    # no fma contraction there (HIP's __fmul_rn/__fadd_rn are plain operators); the MFMA conv
    # file keeps the default.
    extra=(-ffp-contract=off)
    [[ "$s" == "fvp_conv.hip" ]] && extra=()
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "${extra[@]}" -c "${here}/$s" -o "$o" &
  fi
  objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
