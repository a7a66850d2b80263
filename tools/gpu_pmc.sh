cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|GRBM_[A-Z_]+" | sort -u > $out/counters_list.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc1 -o p -- python $root/tools/bench_conv.py --frames 8 --iters 1 > $out/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/pmc2 -o p -- python $root/tools/bench_conv.py --frames 8 --iters 1 > $out/pmc2.log 2>&1
ls -la $out/pmc1 $out/pmc2
