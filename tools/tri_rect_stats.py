#!/usr/bin/env python
"""Rectangle statistics of the fused projection (diagnostics): for the bench workload, the bounding rectangle (in
heatmap pixels) of the taps of every 8 x 4 x BZ voxel block in every view, from the cached sampling coordinates.
    python tools/tri_rect_stats.py [config] [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402

cfgn = sys.argv[1] if len(sys.argv) > 1 else "panoptic"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
cfg = S.make_cfg(cfgn, device=dev, min_score=-1.0)
cams, seq = S.load_cameras(cfgn)
rt = S.resize_transform(cfg).to(dev)
heat = S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100).to(dev)
model = FV.get(cfg).to(dev)
model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
with torch.no_grad():
    model(meta={"seq": [seq] * B}, input_heatmaps=heat, cameras=cams, resize_transform=rt)
e = model.engine
boxes = e.last_jln["boxes"].cpu().numpy()                      # [nP, 9]
grid = e.geo.fine_grid[0]                                         # [V, F0*F1*F2, 2]
F0, F1, F2 = e.fine
V = grid.shape[0]
W, H = e.W, e.H
g = grid.view(V, F0, F1, F2, 2)
for BX, BY, BZ in ((8, 4, 32), (8, 4, 16), (4, 4, 32), (8, 8, 16)):
    areas = []
    for p in range(boxes.shape[0]):
        s0, s1, s2, e0, e1, e2 = boxes[p, 3:9]
        if s0 >= e0 or s1 >= e1 or s2 >= e2:
            continue
        sub = g[:, s0:e0, s1:e1, s2:e2]                        # [V, nx, ny, nz, 2]
        ix = (sub[..., 0] + 1) * (0.5 * (W - 1))
        iy = (sub[..., 1] + 1) * (0.5 * (H - 1))
        x0, y0 = torch.floor(ix), torch.floor(iy)
        inside = ((x0 >= -1) & (x0 < W) & (y0 >= -1) & (y0 < H))   # some tap inside the image
        lo_x, hi_x = x0.clamp(min=0), (x0 + 1).clamp(max=W - 1)
        lo_y, hi_y = y0.clamp(min=0), (y0 + 1).clamp(max=H - 1)
        big = 1e9
        lo_x = torch.where(inside, lo_x, torch.full_like(lo_x, big)); lo_y = torch.where(inside, lo_y, torch.full_like(lo_y, big))
        hi_x = torch.where(inside, hi_x, torch.full_like(hi_x, -big)); hi_y = torch.where(inside, hi_y, torch.full_like(hi_y, -big))
        nx, ny, nz = sub.shape[1:4]
        for xb in range(0, nx, BX):
            for yb in range(0, ny, BY):
                for zb in range(0, nz, BZ):
                    sl = (slice(None), slice(xb, xb + BX), slice(yb, yb + BY), slice(zb, zb + BZ))
                    mnx = lo_x[sl].flatten(1).min(1)[0]; mxx = hi_x[sl].flatten(1).max(1)[0]
                    mny = lo_y[sl].flatten(1).min(1)[0]; mxy = hi_y[sl].flatten(1).max(1)[0]
                    anyv = mxx > -1e8
                    w = (mxx - mnx + 1).clamp(min=0); h = (mxy - mny + 1).clamp(min=0)
                    pitch = (w.long() | 1).float()
                    areas.append(torch.where(anyv, h * pitch, torch.zeros_like(h)))
    a = torch.cat(areas)
    a = a[a > 0]
    print(f"{cfgn} block {BX}x{BY}x{BZ}: {a.numel()} (block, view) rectangles, px: mean {a.mean():.0f} median {a.median():.0f} "
          f"p90 {a.quantile(0.9):.0f} max {a.max():.0f}; <=576: {100 * (a <= 576).float().mean():.1f} %  <=1152: "
          f"{100 * (a <= 1152).float().mean():.1f} %  <=460: {100 * (a <= 460).float().mean():.1f} %  total staged px {a.sum() / 1e6:.1f} M")
