#!/usr/bin/env python
"""Yardstick for the hot path's conv kernels (diagnostics only): the vendor library (MIOpen through torch's conv2d, fp32,
benchmark mode = its fastest algorithm) on P2PNet's layer shapes at B = 8 (240 planes), against the per-op times of
tools/bench_conv.py (profiles/r06_conv_per_op_p2pnet_b8.log).  FLOPs are the direct-conv 2*MAC count, as everywhere."""
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
shapes = [("7x7 15->16 @64x64", 15, 16, 7, 64), ("3x3 16->32 @64x64", 16, 32, 3, 64), ("3x3 32->32 @64x64", 32, 32, 3, 64),
          ("3x3 32->64 @32x32", 32, 64, 3, 32), ("3x3 64->64 @32x32", 64, 64, 3, 32), ("3x3 64->128 @16x16", 64, 128, 3, 16),
          ("3x3 128->128 @16x16", 128, 128, 3, 16), ("1x1 16->32 @64x64", 16, 32, 1, 64)]
N = 240
for name, ci, co, k, hw in shapes:
    x = torch.randn(N, ci, hw, hw, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda")
    for _ in range(5):
        y = F.conv2d(x, w, padding=k // 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = F.conv2d(x, w, padding=k // 2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    fl = 2.0 * N * ci * co * k * k * hw * hw
    print(f"{name:24s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (direct-conv FLOPs)")
