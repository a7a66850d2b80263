run() { echo "### $*"; env "$@" python tools/scratch/dbg_e2e.py 2>&1 | grep -v amdgpu | grep -c "fused False"; }
run A=1
run FVP_CONV_NO_WINO=1
run FVP_WINO_HALF=2
run FVP_WINO_NO_RESW=1
run FVP_WINO_HALF=2 FVP_WINO_NO_RESW=1
run FVP_TRIPLANE_GATHER=1
