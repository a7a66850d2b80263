# PMC passes over the whole hot path (bench.py, B=8, 3 steps).  Separate passes per counter group
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2).  Usage: gpurun -- 'bash tools/gpu_pmc.sh tag'
tag="${1:-r01}"
cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
cmd="python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-mpjpe --no-extra --streams 1"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${tag}_$name -o p -- $cmd > $out/pmc_${tag}_$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
run tcc1 TCC_HIT TCC_MISS TCC_ATOMIC TCC_READ
run fetch FETCH_SIZE
run write WRITE_SIZE
