"""Whole-space back-projection layer of the HDN -- drop-in for the reference's
``lib/models/project_whole.py`` (``ProjectLayer`` :13, ``forward`` :62-88).

Same constructor (``cfg`` only), same forward signature and result
(``cubes [B,J,X,Y,Z]``); the arithmetic is the HIP kernel ``fvp_project_whole``.
"""
import torch
import torch.nn as nn

from ..engine import HotPath


class ProjectLayer(nn.Module):
    def __init__(self, cfg, _engine=None):
        super().__init__()
        self.engine = _engine if _engine is not None else HotPath(cfg)
        self.device = torch.device(cfg.DEVICE)
        self.image_size = cfg.DATASET.IMAGE_SIZE
        self.heatmap_size = cfg.DATASET.HEATMAP_SIZE
        self.ori_image_size = cfg.DATASET.ORI_IMAGE_SIZE
        self.space_size = cfg.CAPTURE_SPEC.SPACE_SIZE
        self.space_center = cfg.CAPTURE_SPEC.SPACE_CENTER
        self.voxels_per_axis = cfg.CAPTURE_SPEC.VOXELS_PER_AXIS
        # reference attribute: per-sequence sampling grids [V,1,nbins,2] (project_whole.py:26,80);
        # filled on first use of a sequence.  The kernels recompute these coordinates on the fly
        # from <= 1 KB of camera parameters, so the cache is informational.
        self.sample_grid = {}

    @property
    def grid(self):
        """[X*Y*Z, 3] voxel centres (project_whole.py:28-47), x slowest / z fastest."""
        ax = self.engine.whole_axes
        mx, my, mz = torch.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
        return torch.stack([mx.reshape(-1), my.reshape(-1), mz.reshape(-1)], dim=1)

    def forward(self, heatmaps, meta, cameras, resize_transform):
        cubes, _ = self.engine.project_whole(heatmaps, meta, cameras, resize_transform, True, False)
        V = heatmaps.shape[1]
        for seq in dict.fromkeys(meta["seq"]):
            if seq not in self.sample_grid:
                self.sample_grid[seq] = self.engine.sample_grid(self.engine.whole_axes, seq, resize_transform, V)
        return cubes
