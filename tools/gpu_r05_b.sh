#!/usr/bin/env bash
# round-5 visit B: parity of the rewritten Winograd kernel + same-box A/B against the round-4 library
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
tag="${1:-r05b}"
echo "== parity"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "${KEXPR:-golden_case or conv_stacks or batch_invariance or seed_sweep or pipelined or hipgraph or register_direct or empty or two_sequences or new_sequence}" > $out/${tag}_pytest.log 2>&1; tail -5 $out/${tag}_pytest.log | cut -c1-300
run() { echo "-- ${1:-current}"; FVP_LIB=$1 timeout 200 python tools/bench_conv.py --net conv_net --frames ${FRAMES:-8} --iters 10 2>&1 | grep -E "k3x3|total" | cut -c1-70; }
run "" | tee $out/${tag}_p2p_new.log
for v in ${VARIANTS:-r04}; do run tools/scratch/libfvp_hip_$v.so | tee $out/${tag}_p2p_$v.log; done
echo "== pipe"
for v in "" ${VARIANTS:-r04}; do lib=""; [ -n "$v" ] && lib=tools/scratch/libfvp_hip_$v.so; FVP_LIB=$lib timeout 200 python tools/bench_pipe.py --config panoptic --batch 8 --streams 4 --steps 100 2>&1 | tail -1; done | tee $out/${tag}_pipe.log
