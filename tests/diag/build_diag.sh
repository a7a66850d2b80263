#!/usr/bin/env bash
# Diagnostics build of the HIP library (TEST / TOOL INFRASTRUCTURE ONLY): the same sources with -DFVP_DIAG=1, in which
# the FVP_* environment switches (kernel selection, tuning, ablations - DESIGN.md section 6) are honoured.  The shipped
# libfvp_hip.so reads no environment variable; the package never loads this file.
#   tests/diag/build_diag.sh  ->  tests/diag/libfvp_hip_diag.so
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
"$root/tools/build_variant.sh" diag -DFVP_DIAG=1 "$@"
cp "$root/tools/scratch/libfvp_hip_diag.so" "$here/libfvp_hip_diag.so"
echo "built $here/libfvp_hip_diag.so"
