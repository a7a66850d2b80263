#!/usr/bin/env python
"""Throughput of the bf16 Pose-ResNet-50 backbone at the Panoptic image size (diagnostics):
N = frames x views images of 512 x 960, HIP-event timed."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import faster_voxelpose_amd.synthetic as S  # noqa: E402
from faster_voxelpose_amd.core import config as CFG  # noqa: E402
from faster_voxelpose_amd.models import resnet as RN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=10)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--h", type=int, default=512)
    ap.add_argument("--w", type=int, default=960)
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
    plan = m._plan(a.h, a.w)
    fl = 0.0
    for op in plan["ops"]:
        if op.kind != 1:
            fl += 2.0 * op.cin * op.cout * (4 if op.kind == 2 else op.kh * op.kw) * op.oh * op.ow
    print(f"{a.images} images {a.h}x{a.w}: {ms:.2f} ms/pass = {ms / a.images:.3f} ms/image, {fl / 1e9:.1f} GFLOP/image, "
          f"{fl * a.images / ms / 1e9:.1f} TFLOP/s (bf16 dense peak 2500)")


if __name__ == "__main__":
    main()
