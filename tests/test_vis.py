"""Visualisation counterpart of lib/utils/vis.py.  CPU only: figures are written from the
oracle's outputs of a small conditioned frame; the projection helper is checked against the oracle's camera model."""
import os

import numpy as np
import torch

import fvp_oracle as O
import fvp_synthetic as S
from faster_voxelpose_amd.core import config as CFG
from faster_voxelpose_amd.utils import vis as VIS


def test_projection_helper_matches_the_oracle_camera_model():
    cams, seq = S.load_cameras("panoptic")
    pts = torch.from_numpy(np.random.default_rng(0).uniform(-1500, 1500, (64, 3)).astype(np.float32)) + \
        torch.tensor([0.0, -500.0, 900.0])
    for cam in cams[seq]:
        want = O.project_points(pts, cam).numpy()
        got = VIS.project_pose_np(pts.numpy(), cam)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-2)       # float64 host helper vs the fp32 chain (pixels)


def test_all_three_views_are_written(tmp_path):
    cfg = S.make_cfg("panoptic", min_score=0.387)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from cases import CONDITIONED
    heat = S.heatmaps_people(cfg, cams, seq, 2, [3, 2], seed=6, **CONDITIONED)
    meta = {"seq": [seq, seq]}
    sd = S.fill_state_dict_conditioned(O.reference_state_dict_shapes(cfg), seed=7)
    fused, planes, centers = O.Oracle(cfg, sd).forward(heat, meta, cams, rt)
    tree = CFG.default_config()
    tree.CAPTURE_SPEC.MIN_SCORE = 0.387
    tree.TEST.VIS_TYPE = ["2d_planes", "image_with_poses", "heatmaps"]
    tree.DATASET.TEST_HEATMAP_SRC = "image"
    Wi, Hi = 240, 128                                                       # small stand-in images
    images = torch.rand(2, 5, 3, Hi, Wi)
    rt_small = torch.as_tensor(rt) * torch.tensor([[0.25], [0.25]])         # image scaled by 1/4 like the heatmaps
    prefix = str(tmp_path / "validation" / "val_00000000")
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    VIS.test_vis_all(tree, meta, cams, rt_small, images, heat, fused, planes, centers, prefix)
    d = os.path.dirname(prefix)
    assert os.path.getsize(os.path.join(d, "2d_planes", "val_00000000.png")) > 10_000
    for c in range(1, 6):
        assert os.path.getsize(os.path.join(d, "image_with_poses", f"val_00000000_view_{c}.jpg")) > 2_000
        assert os.path.getsize(os.path.join(d, "heatmaps", f"val_00000000_view_{c}.jpg")) > 2_000
    # ground truth overlays and the error the reference raises for the 'pred' heatmap source
    meta_gt = dict(meta, num_person=[1, 1], joints_3d=fused[:, :1, :, :3], joints_3d_vis=torch.ones(2, 1, 15))
    VIS.save_2d_planes(tree, meta_gt, fused, planes, centers, prefix + "_gt")
    tree.DATASET.TEST_HEATMAP_SRC = "pred"
    try:
        VIS.test_vis_all(tree, meta, cams, rt_small, images, heat, fused, planes, centers, prefix)
        assert False, "expected ValueError"
    except ValueError as e:
        assert "2D predictions" in str(e)
