#!/usr/bin/env python
"""Per-workgroup timeline of one conv launch (diagnostics): s_memtime stamps written by k_conv_dma
when FVP_CONV_DBG_PTR points at a device buffer."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import faster_voxelpose_amd.synthetic as S  # noqa: E402
from faster_voxelpose_amd import _capi as capi  # noqa: E402
from faster_voxelpose_amd.engine import _ptr  # noqa: E402
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402

op_i = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
model = FV.get(cfg).to(dev)
model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
e = model.engine
model.joint_net.conv_net.ensure_packed()
spec = e.specs["conv_net"]
planes = frames * 30
bufs = [torch.rand((planes,) + tuple(b), device=dev) for b in spec.bufs]
arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
one = (capi.FvpConvOp * 1)(spec.op_array[op_i])
dbg = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
lib = e.lib
for it in range(3):
    if it == 2:
        os.environ["FVP_CONV_DBG_PTR"] = str(dbg.data_ptr())
    capi.check(lib, lib.fvp_conv_stack_run(one, 1, _ptr(e.params["conv_net"]), arr, len(bufs), planes, None, 1, e.stream()), "run")
torch.cuda.synchronize()
d = dbg.cpu().numpy()
n = int((d[:, 0] != 0).sum())
d = d[:n]
hw = d[:, 14].copy(); xcc = d[:, 15].copy(); d[:, 14:] = 0
nst = int((d[0] != 0).sum())
t0 = d[:, 0].min()
print(f"op{op_i}: {n} workgroups, {nst} stamps each; s_memtime ticks (100 MHz constant clock => 10 ns) unless shader clock")
rel = d[:, :nst] - d[:, :1]
print("mean ticks since WG start per stamp:", np.round(rel.mean(0), 1).tolist())
print("mean phase lengths:", np.round(np.diff(rel, axis=1).mean(0), 1).tolist())
start = d[:, 0] - t0
end = d[:, nst - 1] - t0
order = np.argsort(start)
print("WG start times (ticks) quantiles:", np.percentile(start, [0, 10, 25, 50, 75, 90, 100]).round(0).tolist())
print("WG end times quantiles:", np.percentile(end, [0, 10, 25, 50, 75, 90, 100]).round(0).tolist())
print("WG lifetime mean/min/max:", (end - start).mean().round(1), (end - start).min(), (end - start).max())

# placement: HW_ID bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]... (gfx9 layout)
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
x = xcc & 0xf
key = x * 10000 + se * 1000 + sh * 100 + cu
print("distinct (xcc,se,sh,cu) keys:", len(np.unique(key)))
for b in (0, 1, 2, 8, 255, 256, 257, 264, 511, 512, 513):
    if b < n:
        same = np.nonzero(key == key[b])[0][:8]
        print(f"block {b}: xcc {x[b]} se {se[b]} sh {sh[b]} cu {cu[b]}  first blocks on the same CU: {same.tolist()}")
