#!/usr/bin/env bash
# Build a diagnostics variant of the HIP library (always -DFVP_DIAG=1: FVP_* environment switches are honoured) with
# extra compiler flags (e.g. -DFVP_WINO_PK=0):
#   tools/build_variant.sh NAME [flags...]   ->  tools/scratch/libfvp_hip_NAME.so
# Only the tools/bench_*.py scripts (FVP_LIB=...) and tests/diag load such a variant; the product always loads libfvp_hip.so.
# (Environment switches are honoured only by variants built with -DFVP_DIAG=1: tests/diag/build_diag.sh.)
set -euo pipefail
name="$1"; shift
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
csrc="$root/faster-voxelpose_amd/csrc"
out="$root/tools/scratch"; mkdir -p "$out/obj_$name"
objs=()
for s in fvp_capi fvp_project fvp_conv fvp_conv_wino fvp_conv1d_fused fvp_proposal fvp_joint fvp_heatmap fvp_backbone; do
  o="$out/obj_$name/$s.o"
  extra=(-ffp-contract=off); [[ "$s" == "fvp_conv" ]] && extra=(-Wno-inline-asm); [[ "$s" == "fvp_conv_wino" ]] && extra=(-Wno-inline-asm -fno-slp-vectorize)
  if [[ "$s" == "fvp_capi" || "$s" == "fvp_conv" || "$s" == "fvp_conv_wino" || "$s" == "fvp_conv1d_fused" || "$s" == "fvp_joint" || "$s" == "fvp_project" || "$s" == "fvp_backbone" || ! -f "$csrc/$s.o" ]]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFVP_DIAG=1 "${extra[@]}" "$@" -c "$csrc/$s.hip" -o "$o" &
  else
    cp "$csrc/$s.o" "$o"
  fi
  objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libfvp_hip_$name.so"
echo "built $out/libfvp_hip_$name.so"
