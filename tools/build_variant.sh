#!/usr/bin/env bash
# Build a diagnostics variant of the HIP library with extra compiler flags (e.g. -DFVP_WINO_PK=0):
#   tools/build_variant.sh NAME [flags...]   ->  tools/scratch/libfvp_hip_NAME.so
# Only tools/bench_conv.py (FVP_LIB=...) loads such a variant; the product always loads libfvp_hip.so.
set -euo pipefail
name="$1"; shift
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
csrc="$root/faster-voxelpose_amd/csrc"
out="$root/tools/scratch"; mkdir -p "$out/obj_$name"
objs=()
for s in fvp_capi fvp_project fvp_conv fvp_conv1d_fused fvp_proposal fvp_joint fvp_heatmap fvp_backbone; do
  o="$out/obj_$name/$s.o"
  extra=(-ffp-contract=off); [[ "$s" == "fvp_conv" ]] && extra=()
  if [[ "$s" == "fvp_conv" || "$s" == "fvp_conv1d_fused" || "$s" == "fvp_project" || "$s" == "fvp_backbone" || ! -f "$csrc/$s.o" ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "${extra[@]}" "$@" -c "$csrc/$s.hip" -o "$o" &
  else
    cp "$csrc/$s.o" "$o"
  fi
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libfvp_hip_$name.so"
echo "built $out/libfvp_hip_$name.so"
