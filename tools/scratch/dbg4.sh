run() { echo "### $*"; env "$@" python tools/scratch/dbg_e2e.py 2>&1 | grep -v amdgpu | grep -c "fused False"; }
run A=1
run FVP_LIB=tools/scratch/libfvp_hip_scalar.so
run FVP_LIB=tools/scratch/libfvp_hip_scalar.so B=2
