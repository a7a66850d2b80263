run() { echo "### $*"; env "$@" python tools/scratch/dbg_e2e.py 2>&1 | grep -v amdgpu | grep -c "fused False"; }
run A=1
run FVP_WINO_HALF=1
run FVP_BB_NO_BIG=1
run FVP_WINO_LDS_KB=152 FVP_WINO_HALF_PAD=1
