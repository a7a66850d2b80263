run() { echo "### $*"; env "$@" python tools/bench_conv.py --net conv_net --frames 8 --iters 10 --ops 4,10,16 2>&1 | grep -v "amdgpu.ids\|^conv_net"; }
run FVP_CONV_ABLATE=9
run FVP_CONV_ABLATE=25
run FVP_CONV_ABLATE=41
run FVP_CONV_ABLATE=57
run FVP_CONV_ABLATE=73
run FVP_CONV_ABLATE=121
