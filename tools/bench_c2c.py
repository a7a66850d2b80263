#!/usr/bin/env python
"""Time the fused C2CNet kernel (k_conv1d_fused) on N proposal columns (diagnostics; FVP_LIB selects a variant)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd import _capi as capi  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lib  # noqa: E402  (tools/_lib.py: FVP_LIB variant, or the diagnostics build when FVP_* knobs are set)
_lib.select(capi)
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402

dev = "cuda:0"
cfg = S.make_cfg("panoptic", device=dev)
model = FV.get(cfg).to(dev)
model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
for n in (int(a) for a in (sys.argv[1:] or ["80", "10"])):
    z = torch.rand(n, cfg.DATASET.NUM_JOINTS, cfg.CAPTURE_SPEC.VOXELS_PER_AXIS[2], device=dev)
    with torch.no_grad():
        for _ in range(3):
            y = model.pose_net.c2c_net(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = model.pose_net.c2c_net(z)
        e1.record()
        torch.cuda.synchronize()
    print(f"c2c_net {n} columns: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {float(y.double().sum()):.6f}")
