cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4
python - <<'PY'
import sys, time, torch, numpy as np
sys.path[:0] = [".", "tests/golden"]
from heatmap_cases import make_pred2d
from faster_voxelpose_amd.dataset import generate_input_heatmaps
import ctypes as C
from faster_voxelpose_amd import _capi as capi
cfg, preds, rt, sigma = make_pred2d("hm_panoptic_p6"); cfg.DEVICE = "cuda:0"
B = 8
frames = [preds] * B
hm = generate_input_heatmaps(frames, rt, cfg, sigma=sigma); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): hm = generate_input_heatmaps(frames, rt, cfg, sigma=sigma)
torch.cuda.synchronize(); print("end-to-end (host packing + kernel) ms per 8 frames", (time.perf_counter() - t0) / 20 * 1e3)
# kernel only
lib = capi.load()
J = cfg.DATASET.NUM_JOINTS; W, H = cfg.DATASET.HEATMAP_SIZE; V = 5; P = 6
jd = torch.rand(B * V, P, J, 2, dtype=torch.float64, device="cuda") * torch.tensor([960., 512.], dtype=torch.float64, device="cuda")
cd = torch.full((B * V,), P, dtype=torch.int32, device="cuda")
out = torch.empty(B * V, J, H, W, device="cuda"); cl = torch.empty(B * V, H * W, 16, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(11):
    if it == 1: a.record()
    lib.fvp_rasterise_heatmaps(C.c_void_p(jd.data_ptr()), C.c_void_p(cd.data_ptr()), B * V, P, J, W, H, 4.0, 4.0, 3.0, C.c_void_p(out.data_ptr()), C.c_void_p(cl.data_ptr()), 16, s)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 100
print("kernel us per 8 frames (nchw + channels-last outputs)", us, "GB/s written", (out.numel() + cl.numel()) * 4 / us / 1e3)
PY
) > gpurun_out/diag40.log 2>&1
