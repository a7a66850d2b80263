# Per-class kernel time of one serial B=8 step through the fvp_prof_* hooks (FVP_LIB=<variant .so> to compare builds).
python - <<'PY'
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
import fvp_synthetic as S
from faster_voxelpose_amd import _capi as capi
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); import _lib; _lib.select(capi)
from faster_voxelpose_amd.models import faster_voxelpose as FV
dev="cuda:0"
CFGN = os.environ.get("CFG", "panoptic")
cfg = S.make_cfg(CFGN, device=dev, min_score=-1.0)
cams, seq = S.load_cameras(CFGN); rt = S.resize_transform(cfg).to(dev)
B=int(os.environ.get("B","8"))
heat = S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100).to(dev)
meta={"seq":[seq]*B}
model = FV.get(cfg).to(dev); model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
if os.environ.get("FVP_NO_FINE_CACHE"): model.engine.cache_fine_grid = False
lib = capi.load()
names = {capi.K_PROJECT_WHOLE: "project_whole", capi.K_PROJECT_TRIPLANE: "project_triplane", capi.K_SOFTARGMAX:"softargmax", capi.K_OTHER:"other", capi.K_CONV:"conv", capi.K_CONV_WINO:"wino", capi.K_CONV_WINO_SMALL:"wino_sub_chip"}
with torch.no_grad():
    for _ in range(3): out = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    torch.cuda.synchronize()
    lib.fvp_prof_reset(); lib.fvp_prof_enable(2)
    for _ in range(5): out = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    torch.cuda.synchronize()
lib.fvp_prof_enable(0)
for cls, nm in names.items():
    ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
    lib.fvp_prof_read(cls, C.byref(ms), C.byref(n), C.byref(fl))
    print(f"{nm:18s} {ms.value/5*1e3:9.1f} us/step  launches/step {n.value/5:.0f}")
print("planes checksum", float(model.engine.last_jln["planes"].double().sum()), "fused", float(out[0][...,:3].double().abs().sum()))
PY
