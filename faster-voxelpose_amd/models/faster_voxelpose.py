"""Top module -- drop-in for the reference's ``lib/models/faster_voxelpose.py``
(``FasterVoxelPoseNet`` :18-105, ``get`` :108), inference branch.

``forward(backbone=None, views=None, meta=None, targets=None, input_heatmaps=None,
cameras=None, resize_transform=None)`` returns
``(fused_poses [B,N,J,5], plane_poses [3,B,N,J,2], proposal_centers [B,N,7], input_heatmaps, None)``.
"""
import torch
import torch.nn as nn

from ..engine import HotPath
from .human_detection_net import HumanDetectionNet
from .joint_localization_net import JointLocalizationNet


class FasterVoxelPoseNet(nn.Module):
    def __init__(self, cfg, _lib=None):
        super().__init__()
        self.max_people = cfg.CAPTURE_SPEC.MAX_PEOPLE
        self.num_joints = cfg.DATASET.NUM_JOINTS
        self.device = torch.device(cfg.DEVICE)
        self.engine = HotPath(cfg, _lib=_lib)
        self.pose_net = HumanDetectionNet(cfg, _engine=self.engine)
        self.joint_net = JointLocalizationNet(cfg, _engine=self.engine)
        self.lambda_loss_2d = cfg.TRAIN.LAMBDA_LOSS_2D
        self.lambda_loss_1d = cfg.TRAIN.LAMBDA_LOSS_1D
        self.lambda_loss_bbox = cfg.TRAIN.LAMBDA_LOSS_BBOX
        self.lambda_loss_fused = cfg.TRAIN.LAMBDA_LOSS_FUSED
        self.eval()

    def forward(self, backbone=None, views=None, meta=None, targets=None, input_heatmaps=None, cameras=None,
                resize_transform=None):
        if self.training:
            raise NotImplementedError("only the inference branch of FasterVoxelPoseNet.forward is implemented "
                                      "(call model.eval()); training losses are outside the hot path")
        if views is not None:
            # per-view backbone passes, as the reference (:36-38); the backbone is a separate module
            num_views = views.shape[1]
            input_heatmaps = torch.stack([backbone(views[:, c]) for c in range(num_views)], dim=1)
        _, _, proposal_centers, _ = self.pose_net(input_heatmaps, meta, cameras, resize_transform)
        mask = proposal_centers[:, :, 3] >= 0
        fused_poses, plane_poses = self.joint_net.forward5(meta, input_heatmaps, proposal_centers, mask, cameras,
                                                           resize_transform, _reuse_staging=True)
        return fused_poses, plane_poses, proposal_centers, input_heatmaps, None


def get(cfg):
    return FasterVoxelPoseNet(cfg)
