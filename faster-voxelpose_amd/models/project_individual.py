"""Per-person back-projection layer of the JLN -- drop-in for the reference's
``lib/models/project_individual.py`` (``ProjectLayer`` :13, ``forward`` :96-136).

``forward(heatmaps, index, meta, proposal_centers[P,7], cameras, resize_transform)`` returns
``(cubes [P,J,C,C,C], offset [P,3])`` like the reference; window arithmetic
(``fvp_person_boxes``) and sampling (``fvp_project_individual``) are HIP kernels.  The
reference's 164 MB per-sequence fine-grid cache is not needed: sampling coordinates are
recomputed per voxel from the camera parameters.
"""
import torch
import torch.nn as nn

from .. import _capi as capi
from ..engine import HotPath, _ptr
import ctypes as C


class ProjectLayer(nn.Module):
    def __init__(self, cfg, _engine=None):
        super().__init__()
        self.engine = _engine if _engine is not None else HotPath(cfg)
        e = self.engine
        self.device = torch.device(cfg.DEVICE)
        self.image_size = cfg.DATASET.IMAGE_SIZE
        self.heatmap_size = cfg.DATASET.HEATMAP_SIZE
        self.ori_image_size = cfg.DATASET.ORI_IMAGE_SIZE
        dev = self.device
        # the reference's public constants (project_individual.py:22-30,37)
        self.whole_space_center = torch.tensor(cfg.CAPTURE_SPEC.SPACE_CENTER, device=dev)
        self.whole_space_size = torch.tensor(cfg.CAPTURE_SPEC.SPACE_SIZE, device=dev)
        self.ind_space_size = torch.tensor(cfg.INDIVIDUAL_SPEC.SPACE_SIZE, device=dev)
        self.voxels_per_axis = torch.tensor(cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS, device=dev, dtype=torch.int32)
        self.fine_voxels_per_axis = torch.tensor(e.fine, device=dev, dtype=torch.int32)
        self.scale = e.ind_consts[0:3]
        self.bias = e.ind_consts[3:6]
        self.center_grid = e.center_grid
        self.sample_grid = {}

    def forward(self, heatmaps, index, meta, proposal_centers, cameras, resize_transform):
        e = self.engine
        P = proposal_centers.shape[0]
        V = heatmaps.shape[1]
        g = e.geom(resize_transform)
        g.V = V
        fs = e.frame_sets(meta, cameras, V)
        hcl = e.heat_cl(heatmaps, g)
        e._check_tensor(proposal_centers, "proposal_centers")
        centers = proposal_centers.contiguous()
        boxes = torch.empty((P, 9), dtype=torch.int32, device=e.device)
        offset = torch.empty((P, 3), device=e.device)
        cubes = torch.empty((P, e.J, e.C, e.C, e.C), device=e.device)
        if P == 0:
            return cubes, offset
        pf = torch.full((P,), int(index), dtype=torch.int32, device=e.device)
        s = e.stream()
        e._call("fvp_person_boxes", _ptr(centers), P, _ptr(e.ind_consts), e.fine_cube, _ptr(boxes), _ptr(offset), s)
        fa = e.fine_axes
        e._call("fvp_project_individual", _ptr(hcl), _ptr(e.geo.cams), _ptr(fs), _ptr(pf), None, _ptr(boxes), _ptr(fa[0]),
                _ptr(fa[1]), _ptr(fa[2]), _ptr(e.fine_dev), e.C, P, C.byref(g), _ptr(cubes), s)
        self.last_boxes = boxes
        return cubes, offset
