"""Joint Localization Network -- drop-in for the reference's
``lib/models/joint_localization_net.py`` (``JointLocalizationNet`` :36-99).

``forward(meta, heatmaps, proposal_centers [B,N,7], mask [B,N], cameras, resize_transform)``
returns ``(fused [B,N,J,3], plane_poses [3,B,N,J,2])`` and, like the reference (:98),
writes the JLN confidence into ``proposal_centers[..., 4]`` of valid proposals in place.
All B*N proposal slots are processed by one launch per stage (the reference loops over
frames and people in Python); invalid slots are skipped inside the kernels.
"""
import torch
import torch.nn as nn

from ..engine import HotPath
from .cnns_2d import P2PNet
from .project_individual import ProjectLayer
from .weight_net import WeightNet


class SoftArgmaxLayer(nn.Module):
    """Holds beta (:15-18); the expectation runs in ``fvp_softargmax_weightnet``."""

    def __init__(self, cfg):
        super().__init__()
        self.beta = cfg.NETWORK.BETA


class JointLocalizationNet(nn.Module):
    def __init__(self, cfg, _engine=None):
        super().__init__()
        self.engine = _engine if _engine is not None else HotPath(cfg)
        self.conv_net = P2PNet(cfg.DATASET.NUM_JOINTS, cfg.DATASET.NUM_JOINTS, _engine=self.engine)
        self.weight_net = WeightNet(cfg, _engine=self.engine)
        self.project_layer = ProjectLayer(cfg, _engine=self.engine)
        self.soft_argmax_layer = SoftArgmaxLayer(cfg)
        self.fused_projection = True      # False: materialise cubes (fvp_project_individual + fvp_triplane_max)

    def forward(self, meta, heatmaps, proposal_centers, mask, cameras, resize_transform, _reuse_staging=False):
        fused5, plane_poses = self.forward5(meta, heatmaps, proposal_centers, mask, cameras, resize_transform,
                                            _reuse_staging)
        return fused5[..., :3], plane_poses

    def forward5(self, meta, heatmaps, proposal_centers, mask, cameras, resize_transform, _reuse_staging=False):
        self.conv_net.ensure_packed()
        self.weight_net.ensure_packed()
        return self.engine.jln(meta, heatmaps, proposal_centers, mask, cameras, resize_transform,
                               fused=self.fused_projection, reuse_staging=_reuse_staging)
