# PMC passes over the projection kernels:  gpurun -- bash tools/gpu_pmc_projection.sh
cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_tri; mkdir -p $out
cat > /tmp/run_tri.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import fvp_synthetic as S
from faster_voxelpose_amd.models import faster_voxelpose as FV
dev="cuda:0"
cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
cams, seq = S.load_cameras("panoptic"); rt = S.resize_transform(cfg).to(dev)
B=8
heat = S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100).to(dev)
meta={"seq":[seq]*B}
model = FV.get(cfg).to(dev); model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
with torch.no_grad():
    for _ in range(3): out = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
torch.cuda.synchronize()
PY
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o p -- python $root/tools/run_step.py > $out/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES
run sq2 SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC
python - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_tri"
for path in sorted(glob.glob(root+"/*/p_counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n=r["Kernel_Name"]
        if "triplane" not in n and "project_whole" not in n: continue
        n=n.split("(")[0][-40:]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in acc:
        print(path.split("/")[-2], n, {k: f"{sum(v)/len(v):.4g}" for k,v in acc[n].items()})
PY
