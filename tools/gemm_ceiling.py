#!/usr/bin/env python
"""What does the vendor library (hipBLASLt through torch.matmul) reach on the GEMM shapes of the backbone's MFMA-bound layers?
A yardstick for k_bb_conv_dma (diagnostics only): M = pixels, N = couts, K = taps * cin, bf16 in, fp32 accumulate."""
import torch

shapes = [("layer3 3x3 256->256 @32x60 x40", 76800, 256, 2304), ("layer2 3x3 128->128 @64x120 x40", 307200, 128, 1152),
          ("layer4 3x3 512->512 @16x30 x40", 19200, 512, 4608), ("deconv 256->256 @64x120 x40 (one parity class)", 307200, 256, 1024),
          ("deconv 2048->256 @16x30 x40 (one class)", 19200, 256, 8192), ("1x1 256->1024 @32x60 x40", 76800, 1024, 256),
          ("1x1 1024->256 @32x60 x40", 76800, 256, 1024), ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name:50s} M {M:7d} N {N:5d} K {K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
