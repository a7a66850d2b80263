# PMC passes (SQ, LDS, TCC) over the backbone conv kernels:  gpurun -- bash tools/gpu_pmc_backbone.sh
cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_bb; rm -rf $out; mkdir -p $out
cmd="python $root/tools/bench_backbone.py --images 40 --iters 2"
run() { name=$1; shift; FVP_BB_DMA_RING=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o p -- $cmd > $out/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
run tcc1 TCC_HIT TCC_MISS TCC_READ TCC_EA0_RDREQ
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_bb"
for path in sorted(glob.glob(root+"/*/p_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(path)):
        n=r["Kernel_Name"]
        if "k_bb_conv" not in n: continue
        key=n.split("(")[0][-40:]+" g"+r["Grid_Size"]
        acc[key][r["Counter_Name"]]+=float(r["Counter_Value"])
        did=(r["Dispatch_Id"],key)
        if did not in seen: seen.add(did); cnt[key]+=1
    print("==", path.split("/")[-2])
    for k in sorted(acc, key=lambda k:-acc[k].get("SQ_BUSY_CYCLES",acc[k].get("TCC_READ",acc[k].get("SQ_WAVES",0)))):
        print(f"{k:60s} n={cnt[k]:3d} "+" ".join(f"{c}={v/cnt[k]:.3g}" for c,v in sorted(acc[k].items())))
PY
