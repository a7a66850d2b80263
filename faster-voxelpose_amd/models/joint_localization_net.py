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
    """Holds beta (:15-18); the expectation runs in ``fvp_softargmax_weightnet`` -- fused inside
    ``JointLocalizationNet.forward``, or standalone through ``forward(x, grids)`` (:20-34)."""

    def __init__(self, cfg, _engine=None, _weight_net=None):
        super().__init__()
        self.beta = cfg.NETWORK.BETA
        self._engine = [_engine]              # in a list: not a sub-module, not in the state_dict
        self._weight_net = [_weight_net]

    def forward(self, x, grids):
        """x [3,P,J,C,C], grids [3,C*C,2] -> (pose [3,P,J,2], confs [P])."""
        eng = self._engine[0]
        if eng is None:
            raise RuntimeError("SoftArgmaxLayer needs the engine of its JointLocalizationNet for a standalone launch")
        if self._weight_net[0] is not None:
            self._weight_net[0].ensure_packed()
        pose, confs, _ = eng.softargmax_weightnet(x, grids)
        return pose, confs


class JointLocalizationNet(nn.Module):
    def __init__(self, cfg, _engine=None):
        super().__init__()
        self.engine = _engine if _engine is not None else HotPath(cfg)
        self.conv_net = P2PNet(cfg.DATASET.NUM_JOINTS, cfg.DATASET.NUM_JOINTS, _engine=self.engine)
        self.weight_net = WeightNet(cfg, _engine=self.engine)
        self.project_layer = ProjectLayer(cfg, _engine=self.engine)
        self.soft_argmax_layer = SoftArgmaxLayer(cfg, _engine=self.engine, _weight_net=self.weight_net)
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
