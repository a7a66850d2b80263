#!/usr/bin/env python
"""Throughput of the bf16 Pose-ResNet-50 backbone at the Panoptic image size (diagnostics):
N = frames x views images of 512 x 960, HIP-event timed."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd.core import config as CFG  # noqa: E402
from faster_voxelpose_amd.models import resnet as RN  # noqa: E402
from faster_voxelpose_amd import _capi as _capi0  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lib  # noqa: E402  (tools/_lib.py: FVP_LIB variant, or the diagnostics build when FVP_* knobs are set)
_lib.select(_capi0)


def per_op(m, x, a):
    import ctypes as C
    from faster_voxelpose_amd import _capi as capi
    N, _, H, W = x.shape
    plan = m._plan(H, W)
    bufs = []
    for name in plan["names"]:
        c, h, w = plan["shapes"][name]
        bufs.append(torch.randn((N, h, w, c), device="cuda").bfloat16())
    arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
    J = m.num_joints
    cl = torch.empty((N, plan["out_hw"][0] * plan["out_hw"][1], 16), device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = 0.0
    ops = plan["tuned"].get(N, plan["ops"])               # the tile configurations fvp_bb_tune picked, if it ran
    # launch groups: ops the library fuses into one kernel are timed together (stem + max-pool; a layer1 bottleneck =
    # [downsample,] conv1, conv2, conv3), everything else one op at a time
    groups, i = [], 0
    while i < len(ops):
        n = 1
        if a.fused_groups:
            if (ops[i].flags & capi.BB_STEM) and i + 1 < len(ops) and ops[i + 1].kind == capi.BB_MAXPOOL:
                n = 2
            else:
                j = i + 1 if (ops[i].kind == 0 and ops[i].kh == 1 and i + 3 < len(ops) and ops[i + 1].src == ops[i].src
                              and ops[i + 3].res == ops[i].dst) else i      # downsample in front of the block
                if (j + 2 < len(ops) and ops[j].kind == 0 and ops[j].kh == 1 and ops[j].cout == 64 and ops[j + 1].kh == 3
                        and ops[j + 1].stride == 1 and ops[j + 1].src == ops[j].dst and ops[j + 2].kh == 1
                        and ops[j + 2].src == ops[j + 1].dst and ops[j + 2].cout == 256):
                    n = j + 3 - i
        groups.append((i, n))
        i += n
    for i, n in groups:
        op = ops[i + n - 1] if n > 1 else ops[i]
        one = (capi.FvpBbOp * n)(*ops[i:i + n])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(4):
            if it == 1:
                e0.record()
            rc = m.lib.fvp_bb_run(one, n, C.c_void_p(m._wblob.data_ptr()), C.c_void_p(m._eblob.data_ptr()), arr, len(bufs), N,
                                  C.c_void_p(cl.data_ptr()), 16, None, s)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 3
        tot += us
        fl = sum(0.0 if o.kind == 1 else 2.0 * o.cin * o.cout * (4 if o.kind == 2 else o.kh * o.kw) * o.oh * o.ow * N
                 for o in ops[i:i + n])
        first = ops[i]
        if n == 1:
            by = 2.0 * N * (op.cinp * op.h * op.w + op.cout * op.oh * op.ow * (2 if op.res >= 0 else 1))
        else:   # fused group: its input once, its output once (the identity residual IS the input)
            by = 2.0 * N * (first.cinp * first.h * first.w + op.cout * op.oh * op.ow)
        if n > 1:
            print(f"  op{i:2d}-{i + n - 1:2d} fused group of {n}: {first.cin:4d}->{op.cout:4d} @{first.h}x{first.w} -> {op.oh}x{op.ow}  "
                  f"{us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.1f} GB/s (algorithmic in + out)")
            continue
        print(f"  op{i:2d} kind {op.kind} cfg {(op.flags >> 8) & 3} {op.cin:4d}->{op.cout:4d} k{op.kh} s{op.stride} @{op.h}x{op.w}  {us:8.1f} us  "
              f"{fl / us / 1e6:7.1f} TF/s  {by / us / 1e3:7.1f} GB/s")
    print(f"  total {tot / 1e3:.2f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=10)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--h", type=int, default=512)
    ap.add_argument("--w", type=int, default=960)
    ap.add_argument("--per-op", action="store_true")
    ap.add_argument("--no-fused-groups", dest="fused_groups", action="store_false",
                    help="--per-op: time every op alone (the fused kernels then never run)")
    a = ap.parse_args()
    cfg = CFG.default_config()
    m = RN.get(cfg).to("cuda:0")
    m.load_state_dict(S.fill_backbone_state_dict(m.state_dict(), seed=3))
    x = torch.rand(a.images, 3, a.h, a.w, device="cuda")
    with torch.no_grad():
        m.forward_channels_last(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            m.forward_channels_last(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if a.per_op:
        per_op(m, x, a)
    plan = m._plan(a.h, a.w)
    fl = 0.0
    for op in plan["ops"]:
        if op.kind != 1:
            fl += 2.0 * op.cin * op.cout * (4 if op.kind == 2 else op.kh * op.kw) * op.oh * op.ow
    print(f"{a.images} images {a.h}x{a.w}: {ms:.2f} ms/pass = {ms / a.images:.3f} ms/image, {fl / 1e9:.1f} GFLOP/image, "
          f"{fl * a.images / ms / 1e9:.1f} TFLOP/s (bf16 dense peak 2500)")


if __name__ == "__main__":
    main()
