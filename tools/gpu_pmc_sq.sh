# SQ counter passes only (wave cycles, waits, VALU / MFMA / LDS activity): gpurun -- 'bash tools/gpu_pmc_sq.sh tag'
tag="${1:-r05}"
cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
cmd="python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-mpjpe --no-extra --streams 1"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${tag}_$name -o p -- $cmd > $out/pmc_${tag}_$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES
cd $root; python tools/pmc_summary.py $tag 2>/dev/null | grep -A1 "k_conv_wino\|k_conv_reg\|k_conv_dma<7" | cut -c1-400
