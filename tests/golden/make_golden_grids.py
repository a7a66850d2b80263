"""Golden digests of the reference's cached sampling grids and of its checkpoint layout (build
container only) -> ``grids.npz``:

  {shape}_whole  : project_whole.ProjectLayer.sample_grid[seq][:, 0, ::stride]      (project_whole.py:75-83)
  {shape}_fine   : project_individual.ProjectLayer.sample_grid[seq][:, ::sx, ::sy, ::sz]  (project_individual.py:82-94)
  {shape}_keys / {shape}_shapes : the reference model's state_dict key list (485 entries) and tensor shapes

for the Panoptic / Shelf / Campus shape sets.  The grids are produced by running the reference's own
layers once on zero heatmaps (the grid depends on cameras and config only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import _refimport as R  # noqa: E402
import fvp_synthetic as S  # noqa: E402

FINE_STRIDE = (23, 23, 7)


def main():
    ref = R.import_reference()
    out = {}
    for shape in ("panoptic", "shelf", "campus"):
        cfg = S.make_cfg(shape, device="cpu", min_score=-1.0)
        cams, seq = S.load_cameras(shape)
        rt = S.resize_transform(cfg)
        w, h = cfg.DATASET.HEATMAP_SIZE
        heat = torch.zeros(1, cfg.DATASET.CAMERA_NUM, cfg.DATASET.NUM_JOINTS, h, w)
        meta = {"seq": [seq]}
        with R.quiet(), torch.no_grad():
            model = ref.faster_voxelpose.get(cfg).eval()
            model.pose_net.project_layer(heat, meta, cams, rt)
            pc = torch.zeros(1, 7)
            pc[0, :3] = torch.tensor(cfg.CAPTURE_SPEC.SPACE_CENTER)
            pc[0, 5:7] = 0.5
            model.joint_net.project_layer(heat, 0, meta, pc, cams, rt)
        whole = model.pose_net.project_layer.sample_grid[seq].squeeze(1)                 # [V, n, 2]
        fine = model.joint_net.project_layer.sample_grid[seq]                            # [V, fx, fy, fz, 2]
        assert not torch.isnan(whole).any() and not torch.isnan(fine).any()
        ws = max(1, whole.shape[1] // 1500)
        out[f"{shape}_whole"] = whole[:, ::ws].numpy()
        out[f"{shape}_whole_stride"] = np.int64(ws)
        sx, sy, sz = FINE_STRIDE
        out[f"{shape}_fine"] = fine[:, ::sx, ::sy, ::sz].numpy()
        out[f"{shape}_fine_dims"] = np.array(fine.shape[1:4], np.int64)
        sd = model.state_dict()
        out[f"{shape}_keys"] = np.array(list(sd.keys()))
        out[f"{shape}_shapes"] = np.array([",".join(str(int(d)) for d in v.shape) for v in sd.values()])
        print(shape, "whole", tuple(whole.shape), "fine", tuple(fine.shape), "keys", len(sd))
    out["fine_stride"] = np.array(FINE_STRIDE, np.int64)
    np.savez_compressed(os.path.join(HERE, "grids.npz"), **out)
    print("wrote grids.npz", os.path.getsize(os.path.join(HERE, "grids.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
