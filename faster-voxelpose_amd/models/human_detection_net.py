"""Human Detection Network -- drop-in for the reference's
``lib/models/human_detection_net.py`` (``HumanDetectionNet`` :67-104, inference branch).

``forward(heatmaps [B,V,J,H,W], meta, cameras, resize_transform)`` returns
``(hm2d [B,1,X,Y], hm1d [B,N,Z], proposal_centers [B,N,7], bbox_preds [B,X*Y,2])``.
Everything between runs as HIP kernels on the current stream without a host
synchronisation: back-projection with fused z-max, CenterNet, NMS/top-k, gathers, C2CNet,
z arg-max and proposal packing.  The training-only ground-truth matching
(``ProposalLayer.filter_proposal`` :25-42) is out of scope.
"""
import torch
import torch.nn as nn

from ..engine import HotPath
from .cnns_1d import C2CNet
from .cnns_2d import CenterNet
from .project_whole import ProjectLayer


class ProposalLayer(nn.Module):
    """Drop-in for the reference's ProposalLayer (:13-65, eval branch).  Inside ``HumanDetectionNet.forward`` the packing
    is fused with the z arg-max (``fvp_proposals``); ``forward`` here is the same arithmetic as a standalone launch
    (``fvp_proposal_layer``) for callers that use the layer on its own."""

    def __init__(self, cfg, _engine=None):
        super().__init__()
        self.max_people = cfg.CAPTURE_SPEC.MAX_PEOPLE
        self.min_score = cfg.CAPTURE_SPEC.MIN_SCORE
        self.device = torch.device(cfg.DEVICE)
        e = _engine
        self.engine = e
        self.scale = e.prop_sb[0:3]
        self.bias = e.prop_sb[3:6]

    def forward(self, topk_index, topk_confs, match_bbox_preds, meta):
        """``topk_index`` [B,N,3] (voxel indices, any integer dtype), ``topk_confs`` [B,N], ``match_bbox_preds`` [B,N,2]
        -> ``proposal_centers`` [B,N,7] (human_detection_net.py:44-65)."""
        if self.training and ("roots_3d" in meta and "num_person" in meta):
            raise NotImplementedError("training-time proposal matching is outside the inference hot path")
        return self.engine.proposal_layer(topk_index, topk_confs, match_bbox_preds, self.min_score)


class HumanDetectionNet(nn.Module):
    def __init__(self, cfg, _engine=None):
        super().__init__()
        self.engine = _engine if _engine is not None else HotPath(cfg)
        self.max_people = cfg.CAPTURE_SPEC.MAX_PEOPLE
        self.project_layer = ProjectLayer(cfg, _engine=self.engine)
        self.center_net = CenterNet(cfg.DATASET.NUM_JOINTS, 1, _engine=self.engine)
        self.c2c_net = C2CNet(cfg.DATASET.NUM_JOINTS, 1, _engine=self.engine)
        self.proposal_layer = ProposalLayer(cfg, _engine=self.engine)

    def forward(self, heatmaps, meta, cameras, resize_transform):
        if self.training and ("roots_3d" in meta and "num_person" in meta):
            raise NotImplementedError("training-time proposal matching is outside the inference hot path")
        self.center_net.ensure_packed()
        self.c2c_net.ensure_packed()
        return self.engine.hdn(heatmaps, meta, cameras, resize_transform)
