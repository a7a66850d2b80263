#!/usr/bin/env python
"""Prototype: side branches of a conv stack (1x1 skip convs, skip_res blocks, CenterNet's second head) on side streams.
   python tools/bench_branch.py [--frames 1]"""
import argparse
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd import _capi as capi  # noqa: E402
import _lib  # noqa: E402

_lib.select(capi)
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402


def schedule(spec, nside=2):
    ops, n = spec.ops, len(spec.ops)
    cons, prod = defaultdict(list), {}
    for i, o in enumerate(ops):
        cons[o["src"]].append((i, "src"))
        if o["res"] >= 0:
            cons[o["res"]].append((i, "res"))
        prod[o["dst"]] = i
    final = ops[-1]["dst"]
    side = [False] * n
    for i in reversed(range(n)):
        d = ops[i]["dst"]
        c = cons[d]
        side[i] = (d != final) if not c else all((role == "res" and ops[j]["src"] != d) or side[j] for j, role in c)
    chain_of, chains = {}, []
    for i in range(n):
        if not side[i]:
            continue
        p = prod.get(ops[i]["src"], -1)
        if p >= 0 and side[p]:
            chain_of[i] = chain_of[p]
            chains[chain_of[i]]["ops"].append(i)
        else:
            chain_of[i] = len(chains)
            chains.append({"ops": [i]})
    for ch in chains:
        ins = set()
        for i in ch["ops"]:
            for b in (ops[i]["src"], ops[i]["res"]):
                if b >= 0 and prod.get(b, -1) not in ch["ops"]:
                    ins.add(prod.get(b, -1))
        assert all(p < 0 or not side[p] for p in ins)
        ch["ready"] = max(ins)
        outs = [j for i in ch["ops"] for j, _ in cons[ops[i]["dst"]] if not side[j]]
        ch["deadline"] = min(outs) if outs else n
        ch["stream"] = 0 if ch["deadline"] - ch["ready"] <= 4 else min(1, nside - 1)
    cmds, seg = [], []
    main_ops = [i for i in range(n) if not side[i]]

    def flush():
        if seg:
            cmds.append(("run", -1, list(seg)))
            seg.clear()

    def start(ready):
        todo = sorted([c for c in range(len(chains)) if chains[c]["ready"] == ready], key=lambda c: chains[c]["deadline"])
        if todo:
            flush()
            cmds.append(("record", -1, ("fork", ready)))
            for c in todo:
                s = chains[c]["stream"]
                cmds.append(("wait", s, ("fork", ready)))
                cmds.append(("run", s, chains[c]["ops"]))
                cmds.append(("record", s, ("done", c)))
    start(-1)
    for k, m in enumerate(main_ops):
        for c, ch in enumerate(chains):
            if ch["deadline"] == m:
                flush()
                cmds.append(("wait", -1, ("done", c)))
        seg.append(m)
        nxt = main_ops[k + 1] if k + 1 < len(main_ops) else -1
        fusable = nxt >= 0 and ops[nxt]["kind"] == capi.OP_POOL2 and ops[nxt]["src"] == ops[m]["dst"]
        if not fusable:
            # ready points: this op, and a pool just appended whose producer was a ready point too
            for r in ([m] + ([main_ops[k - 1]] if ops[m]["kind"] == capi.OP_POOL2 and k > 0 else [])):
                start(r)
    flush()
    for c, ch in enumerate(chains):
        if ch["deadline"] == n:
            cmds.append(("wait", -1, ("done", c)))
    return cmds, chains


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--nside", type=int, default=2)
    args = ap.parse_args()
    dev = "cuda:0"
    cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
    model = FV.get(cfg).to(dev)
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
    e = model.engine
    for m in (model.pose_net.center_net, model.joint_net.conv_net):
        m.ensure_packed()
    lib = e.lib
    sides = [torch.cuda.Stream() for _ in range(args.nside)]
    for net, planes in (("center_net", args.frames), ("conv_net", args.frames * 30)):
        spec = e.specs[net]
        bufs = [torch.rand((planes,) + tuple(b), device=dev) for b in spec.bufs]
        arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
        params = C.c_void_p(e.params[net].data_ptr())
        cmds, chains = schedule(spec, args.nside)
        if net == "center_net":
            print("chains:", [(c["ops"], c["ready"], c["deadline"], c["stream"]) for c in chains])
        subs = {id(c): (capi.FvpConvOp * len(c[2]))(*[spec.op_array[i] for i in c[2]]) for c in cmds if c[0] == "run"}
        events = {}

        def seq():
            capi.check(lib, lib.fvp_conv_stack_run(spec.op_array, len(spec.ops), params, arr, len(bufs), planes, None, 1,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "run")

        def branched():
            main_s = torch.cuda.current_stream()
            for c in cmds:
                st = main_s if c[1] < 0 else sides[c[1]]
                if c[0] == "run":
                    capi.check(lib, lib.fvp_conv_stack_run(subs[id(c)], len(c[2]), params, arr, len(bufs), planes, None, 1,
                                                           C.c_void_p(st.cuda_stream)), "run")
                elif c[0] == "record":
                    ev = events.setdefault(c[2], torch.cuda.Event())
                    ev.record(st)
                else:
                    st.wait_event(events[c[2]])

        ref = None
        for name, fn in (("sequential", seq), ("branched", branched), ("sequential", seq), ("branched", branched)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(args.iters + 3):
                if it == 3:
                    a.record()
                fn()
            b.record()
            torch.cuda.synchronize()
            outs = [bufs[i].clone() for i in spec.outputs.values()]
            if ref is None:
                ref = outs
            same = all(torch.equal(x, y) for x, y in zip(ref, outs))
            print(f"{net} planes {planes} {name:10s}: {a.elapsed_time(b) * 1e3 / args.iters:8.1f} us   same bits: {same}")


if __name__ == "__main__":
    main()
