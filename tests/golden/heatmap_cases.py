"""Recipes of the rasteriser fixtures (shared by the generator and the tests): seeded synthetic 2-D
detections per view in ORIGINAL image pixels, including people partly / fully outside the image."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import fvp_synthetic as S  # noqa: E402

HEATMAP_CASES = {
    # name -> (shape set, people per view, seed, sigma)
    "hm_shelf_p4": ("shelf", [4, 3, 4, 2, 0], 5, 3),
    "hm_campus_p3": ("campus", [3, 3, 1], 6, 3),
    "hm_panoptic_p6": ("panoptic", [6, 5, 6, 6, 4], 7, 3),
}


def make_pred2d(case):
    shape, people, seed, sigma = HEATMAP_CASES[case]
    cfg = S.make_cfg(shape, device="cpu")
    rng = np.random.default_rng(seed)
    ow, oh = cfg.DATASET.ORI_IMAGE_SIZE
    J = cfg.DATASET.NUM_JOINTS
    all_preds = []
    for v, n in enumerate(people):
        preds = []
        for p in range(n):
            # a person = a box of random size somewhere around the image (some partly outside)
            cx = rng.uniform(-0.1 * ow, 1.1 * ow)
            cy = rng.uniform(-0.1 * oh, 1.1 * oh)
            hgt = rng.uniform(0.08 * oh, 0.9 * oh)
            wid = hgt * rng.uniform(0.25, 0.6)
            pts = np.stack([cx + (rng.random(J) - 0.5) * wid, cy + (rng.random(J) - 0.5) * hgt,
                            rng.random(J)], axis=1)
            preds.append(pts.astype(np.float64))
        all_preds.append(preds)
    rt = S.resize_transform(cfg).numpy().astype(np.float64)
    return cfg, all_preds, rt, sigma
