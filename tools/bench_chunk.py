#!/usr/bin/env python
"""P2PNet over plane groups: does running the stack on G groups of planes/G planes (working set inside the 256 MB
Infinity Cache) beat one pass over all planes?   python tools/bench_chunk.py [--frames 8] [--groups 1,2,3,4,6]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import fvp_synthetic as S

from faster_voxelpose_amd import _capi as capi  # noqa: E402
import _lib  # noqa: E402

_lib.select(capi)
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--groups", default="1,2,3,4,6")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--split", default="", help="a:b[,c:d] - only these op ranges run per group, the rest on all planes")
    args = ap.parse_args()
    dev = "cuda:0"
    cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
    model = FV.get(cfg).to(dev)
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
    e = model.engine
    model.joint_net.conv_net.ensure_packed()
    spec = e.specs["conv_net"]
    planes = args.frames * 30
    bufs = [torch.rand((planes,) + tuple(b), device=dev) for b in spec.bufs]
    lib = e.lib
    params = C.c_void_p(e.params["conv_net"].data_ptr())
    nops = len(spec.ops)

    def run(lo, hi, p0, p1):
        arr = (C.c_void_p * len(bufs))(*[t[p0:p1].data_ptr() for t in bufs])
        sub = (capi.FvpConvOp * (hi - lo))(*[spec.op_array[i] for i in range(lo, hi)])
        capi.check(lib, lib.fvp_conv_stack_run(sub, hi - lo, params, arr, len(bufs), p1 - p0, None, 1, e.stream()), "run")

    for g in [int(x) for x in args.groups.split(",")]:
        step = (planes + g - 1) // g
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(args.iters + 2):
            if it == 2:
                a.record()
            if args.split:
                cur = 0
                for rng in args.split.split(","):
                    lo, hi = (int(x) for x in rng.split(":"))
                    if lo > cur:
                        run(cur, lo, 0, planes)
                    for p0 in range(0, planes, step):
                        run(lo, hi, p0, min(planes, p0 + step))
                    cur = hi
                if cur < nops:
                    run(cur, nops, 0, planes)
            else:
                for p0 in range(0, planes, step):
                    run(0, nops, p0, min(planes, p0 + step))
        b.record()
        torch.cuda.synchronize()
        print(f"groups {g} ({step} planes each){' split ' + args.split if args.split else ''}: {a.elapsed_time(b) * 1e3 / args.iters:8.1f} us per pass")


if __name__ == "__main__":
    main()
