run() { echo "### $*"; env "$@" python tools/bench_conv.py --net conv_net --frames 8 --iters 10 --ops 1,3,4,7,9,10,13,15,16 2>&1 | grep -v "amdgpu.ids\|^conv_net"; }
run A=default
run FVP_CONV_ABLATE=8
run FVP_CONV_ABLATE=9
