#!/usr/bin/env python
"""Where do k_conv_reg and k_conv_dma differ?  (GPU debugging aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from common import reg_stack, run_custom_conv_stack
from faster_voxelpose_amd import _capi as capi
lib = capi.load()
DEV = "cuda:0"
for fused, hc in ((False, 15), (True, 17)):
    spec, w, ref, outs = reg_stack(seed=4, fused_head=fused, head_cout=hc)
    x = torch.from_numpy(np.random.default_rng(6).normal(size=(600, 32, 16, 16)).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    big = run_custom_conv_stack(lib, DEV, spec, w, x, st)
    few = run_custom_conv_stack(lib, DEV, spec, w, x[:3], st)
    for name, o in outs.items():
        a, b = big[o][:3].cpu().numpy(), few[o].cpu().numpy()
        bad = np.argwhere(a != b)
        print(f"fused={fused} {name}: shape {a.shape} mismatches {len(bad)} of {a.size}")
        if len(bad):
            print("   planes", np.unique(bad[:, 0])[:10], "channels", np.unique(bad[:, 1])[:40])
            print("   rows", np.unique(bad[:, 2])[:40], "cols", np.unique(bad[:, 3])[:40])
            i = tuple(bad[0]); print("   first", i, a[i], b[i])
