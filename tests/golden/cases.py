"""Recipes of the golden cases (shared by the generator and the tests).

name -> (shape set, heatmap flavour, batch, people, heatmap seed, weight seed, MIN_SCORE override)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import fvp_synthetic as S  # noqa: E402

CASES = {
    "tiny_g_b2_all": ("tiny", "g", 2, 2, 3, 7, -1.0),
    "tiny_u_b3_thr": ("tiny", "u", 3, 0, 2, 7, 45.0),
    "campus_u_b2_all": ("campus", "u", 2, 0, 2, 7, -1.0),
    "panoptic_g_b1_all": ("panoptic", "g", 1, 4, 3, 7, -1.0),
    "panoptic_g_b2_thr": ("panoptic", "g", 2, 3, 5, 7, 17.9),
    "panoptic_u_b1_all": ("panoptic", "u", 1, 0, 2, 7, -1.0),
    "shelf_g_b1_all": ("shelf", "g", 1, 4, 3, 11, -1.0),
    # BASELINE.json configs[3] shape: 128x128x32 detection grid, jln128
    "panoptic128_g_b1_all": ("panoptic128", "g", 1, 4, 10, 7, -1.0),
}


def make_inputs(case, device="cpu"):
    shape, flavour, B, people, hseed, wseed, ms = CASES[case]
    cfg = S.make_cfg(shape, device=device, min_score=ms)
    cams, seq = S.load_cameras(shape)
    rt = S.resize_transform(cfg)
    if flavour == "g":
        heat = S.heatmaps_blobs(cfg, cams, seq, B, people=people, seed=hseed)
    else:
        heat = S.heatmaps_uniform(cfg, B, seed=hseed)
    meta = {"seq": [seq] * B}
    return cfg, cams, seq, rt, heat, meta, wseed
