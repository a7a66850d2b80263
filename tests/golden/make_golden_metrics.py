"""Golden vectors for the evaluators, from the reference's own ``Panoptic.evaluate`` /
``Shelf.evaluate`` / ``Shelf.coco2shelf3D`` (build container only).  Instances are created
without ``__init__`` and given exactly the attributes those methods read; Shelf's ``actorsGT.mat``
is a synthetic file with the real file's cell-array nesting."""
import os
import re
import sys
import tempfile
import types

import numpy as np
import scipy.io as scio
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), HERE]
import _refimport as R  # noqa: E402


def synth(seed, frames=12, J=15, N=10):
    rng = np.random.default_rng(seed)
    preds, gts, vis = [], [], []
    for _ in range(frames):
        P = int(rng.integers(0, 5))
        g = rng.normal(0, 800, (P, J, 3))
        v = (rng.random((P, J)) > 0.1).astype(np.float64)
        p = np.zeros((N, J, 5))
        p[:, :, 3] = -1
        k = int(rng.integers(0, N + 1))
        for n in range(k):
            src = g[rng.integers(0, P)] if P and rng.random() < 0.8 else rng.normal(0, 800, (J, 3))
            p[n, :, :3] = src + rng.normal(0, rng.choice([10, 40, 120, 400]), (J, 3))
            p[n, :, 3] = 0
            p[n, :, 4] = rng.random()
        preds.append(p)
        gts.append(g)
        vis.append(v)
    return preds, gts, vis


def main():
    R.import_reference()
    sys.modules.setdefault("json_tricks", types.ModuleType("json_tricks"))
    sys.path.insert(0, os.path.join(R.REF_ROOT, "lib"))
    from dataset.panoptic import Panoptic
    from dataset.shelf import Shelf
    out = {}
    # ---- Panoptic
    preds, gts, vis = synth(1)
    ds = object.__new__(Panoptic)
    ds.db = [{"meta": {"num_person": len(g), "joints_3d": g, "joints_3d_vis": v}} for g, v in zip(gts, vis)]
    ds.db_size = len(ds.db)
    metric, msg = ds.evaluate([torch.from_numpy(p) for p in preds])
    nums = [float(x) for x in re.findall(r":\s*([0-9.einf]+)", msg.split("\n", 1)[1])]
    out.update(pan_metric=metric, pan_numbers=np.array(nums))
    print(msg)
    # ---- Shelf (COCO-17 predictions, 14-joint actors)
    rng = np.random.default_rng(2)
    frames, P = 9, 4
    coco_gt = [[rng.normal(0, 600, (17, 3)) + rng.normal(0, 1500, (1, 3)) for _ in range(frames)] for _ in range(P)]
    # actors in Shelf order, metres, with a little annotation noise; some frames without the actor
    actors = [[(Shelf.coco2shelf3D(coco_gt[p][f].copy()) / 1000.0 + rng.normal(0, 0.01, (14, 3))
                if rng.random() < 0.8 else None) for f in range(frames)] for p in range(P)]
    cell = np.empty((1, P), dtype=object)
    for p in range(P):
        col = np.empty((frames, 1), dtype=object)
        for f in range(frames):
            col[f, 0] = actors[p][f] if actors[p][f] is not None else np.zeros((1, 0))
        cell[0, p] = col
    preds17 = []
    for f in range(frames):
        p = np.zeros((10, 17, 5))
        p[:, :, 3] = -1
        n = 0
        for a in range(P):
            if rng.random() < 0.85:
                p[n, :, :3] = coco_gt[a][f] + rng.normal(0, rng.choice([15, 60, 200]), (17, 3))
                p[n, :, 3] = 0
                p[n, :, 4] = rng.random()
                n += 1
        if n == 0:
            p[0, :, :3] = rng.normal(0, 600, (17, 3))
            p[0, :, 3] = 0
        preds17.append(p)
    with tempfile.TemporaryDirectory() as d:
        scio.savemat(os.path.join(d, "actorsGT.mat"), {"actor3D": cell})
        sh = object.__new__(Shelf)
        sh.dataset_dir = d
        sh.frame_range = list(range(frames))
        metric, msg = sh.evaluate([torch.from_numpy(p) for p in preds17])
    print(msg)
    nums = [float(x) for x in re.findall(r"([0-9]+.[0-9]+)", msg.split("\n")[1])]
    out.update(shelf_metric=metric, shelf_numbers=np.array(nums),
               shelf_conv=np.stack([Shelf.coco2shelf3D(p[0, :, :3].copy()) for p in preds17]),
               shelf_preds=np.stack(preds17),
               shelf_actors=np.stack([[a if a is not None else np.full((14, 3), np.nan) for a in row] for row in actors]),
               pan_preds=np.stack(preds), pan_gt_count=np.array([len(g) for g in gts]),
               pan_gt=np.concatenate(gts), pan_vis=np.concatenate(vis))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print({k: (v if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
