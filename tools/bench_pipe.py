#!/usr/bin/env python
"""Pipelined throughput of the voxel path (B frames per batch, D batches in flight) for A/B runs of library variants
(FVP_LIB=<variant .so>, or FVP_* knobs -> the diagnostics build; tools/_lib.py).  Diagnostics only: the judged number
comes from bench.py with the product library."""
import argparse
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # HIP runtime: see bench.py
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd import _capi as capi  # noqa: E402
import _lib  # noqa: E402

_lib.select(capi)
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="panoptic")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--graph", action="store_true", help="one hipGraph per pipeline slot (FV.GraphedPipeline)")
a = ap.parse_args()
dev = "cuda:0"
cfg = S.make_cfg(a.config, device=dev, min_score=-1.0)
cams, seq = S.load_cameras(a.config)
rt = S.resize_transform(cfg).to(dev)
heats = [S.heatmaps_blobs(cfg, cams, seq, a.batch, people=4, seed=100 + i).to(dev) for i in range(4)]
meta = {"seq": [seq] * a.batch}
model = FV.get(cfg).to(dev)
model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
if a.graph:
    gp = FV.GraphedPipeline(model, a.streams, meta, heats[0], cams, rt)

    class _P:                                            # same submit signature as PipelinedForward
        def submit(self, **kw):
            return gp.submit(kw["input_heatmaps"])

        def synchronize(self):
            gp.synchronize()
    pipe = _P()
else:
    pipe = FV.PipelinedForward(model, depth=a.streams)
with torch.no_grad():
    for i in range(6):
        pipe.submit(meta=meta, input_heatmaps=heats[i % 4], cameras=cams, resize_transform=rt)
    pipe.synchronize()
    rates = []
    subs = []
    for r in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            pipe.submit(meta=meta, input_heatmaps=heats[i % 4], cameras=cams, resize_transform=rt)
        t1 = time.perf_counter()
        pipe.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rates.append(a.steps * a.batch / (t2 - t0))
        subs.append((1e3 * (t1 - t0) / a.steps, 1e3 * (t2 - t1)))
print(f"   host submit ms/step, drain ms: " + " ".join(f"{x:.3f}/{y:.2f}" for x, y in subs))
print(f"{os.path.basename(capi.LIB_PATH)}: {a.config} B={a.batch} x{a.streams}{' hipGraph slots' if a.graph else ''}: frames/s " + " ".join(f"{x:.0f}" for x in rates))
