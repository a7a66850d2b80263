#!/usr/bin/env python
"""Per-op timing of a conv stack on the GPU (diagnostics): runs every FvpConvOp of the P2PNet /
CenterNet / C2CNet plan on its own with HIP events and prints time, algorithmic TFLOP/s and
the activation bytes it moves.  Env knobs of the library (FVP_CONV_*) apply."""
import argparse
import ctypes as C
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
from faster_voxelpose_amd.engine import _ptr  # noqa: E402
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="conv_net")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ops", default="", help="comma-separated op indices (default: all)")
    ap.add_argument("--span", default="", help="a:b - additionally time ops a..b-1 as ONE stack run (fusions apply)")
    args = ap.parse_args()
    dev = "cuda:0"
    cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
    model = FV.get(cfg).to(dev)
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
    e = model.engine
    for m in (model.pose_net.center_net, model.pose_net.c2c_net, model.joint_net.conv_net):
        m.ensure_packed()
    spec = e.specs[args.net]
    planes = {"conv_net": args.frames * 30, "center_net": args.frames, "c2c_net": args.frames * 10}[args.net]
    bufs = [torch.rand((planes,) + tuple(b), device=dev) for b in spec.bufs]
    arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
    lib = e.lib
    tot = 0.0
    print(f"{args.net}: {planes} planes   env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FVP_")))
    only = {int(x) for x in args.ops.split(",") if x}
    for i, op in enumerate(spec.ops):
        if only and i not in only:
            continue
        one = (capi.FvpConvOp * 1)(spec.op_array[i])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(args.iters + 1):
            if it == 1:
                a.record()
            capi.check(lib, lib.fvp_conv_stack_run(one, 1, _ptr(e.params[args.net]), arr, len(bufs), planes, None, 1,
                                                   e.stream()), "run")
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / args.iters
        tot += us
        taps = op["kh"] * op["kw"] if op["kind"] == capi.OP_CONV else (4 if op["h"] > 1 else 2)
        fl = 0.0 if op["kind"] == capi.OP_POOL2 else 2.0 * op["cin"] * op["cout"] * taps * op["h"] * op["w"] * planes
        osz = op["h"] * op["w"] * (4 if op["kind"] == capi.OP_CONVT2 and op["h"] > 1 else 1)
        byt = 4.0 * planes * (op["cin"] * op["h"] * op["w"] + op["cout"] * osz * (2 if op["res"] >= 0 else 1))
        kind = {0: "conv", 1: "pool", 2: "convT"}[op["kind"]]
        print(f"  op{i:2d} {kind:5s} {op['cin']:3d}->{op['cout']:3d} k{op['kh']}x{op['kw']} @{op['h']}x{op['w']}  "
              f"{us:8.1f} us  {fl / us / 1e6:6.1f} TF/s  {byt / us / 1e3:7.1f} GB/s")
    print(f"  total {tot:.1f} us per pass = {tot / args.frames:.1f} us/frame")
    if args.span:
        lo, hi = (int(x) for x in args.span.split(":"))
        sub = (capi.FvpConvOp * (hi - lo))(*[spec.op_array[i] for i in range(lo, hi)])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(args.iters + 1):
            if it == 1:
                a.record()
            capi.check(lib, lib.fvp_conv_stack_run(sub, hi - lo, _ptr(e.params[args.net]), arr, len(bufs), planes, None, 1,
                                                   e.stream()), "run")
        b.record()
        torch.cuda.synchronize()
        print(f"  span {lo}:{hi} as one stack  {a.elapsed_time(b) * 1e3 / args.iters:8.1f} us")


if __name__ == "__main__":
    main()
