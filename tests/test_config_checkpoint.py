"""Checkpoint / YAML ingestion.  CPU only."""
import os

import numpy as np
import pytest
import torch

import fvp_synthetic as S
from faster_voxelpose_amd.core import config as CFG
from faster_voxelpose_amd.utils import checkpoint as CK

SHELF_YAML = """
CUDNN: {BENCHMARK: true, DETERMINISTIC: false, ENABLED: true}
BACKBONE: 'resnet'
MODEL: 'faster_voxelpose'
DEVICE: 'cuda:0'
DATASET:
  DATADIR: "data/Shelf"
  COLOR_RGB: True
  TEST_DATASET: 'shelf'
  TEST_HEATMAP_SRC: 'pred'
  CAMERA_NUM: 5
  ORI_IMAGE_SIZE: [1032, 776]
  IMAGE_SIZE: [800, 608]
  HEATMAP_SIZE: [200, 152]
  NUM_JOINTS: 17
  ROOT_JOINT_ID: [11, 12]
NETWORK: {SIGMA: 3, BETA: 100, NUM_CHANNEL_JOINT_FEAT: 32, NUM_CHANNEL_JOINT_HIDDEN: 64}
TEST: {MODEL_FILE: 'model_best.pth.tar', BATCH_SIZE: 16}
CAPTURE_SPEC:
  SPACE_SIZE: [8000.0, 8000.0, 2000.0]
  SPACE_CENTER: [450.0, -320.0, 800.0]
  VOXELS_PER_AXIS: [80, 80, 20]
  MAX_PEOPLE: 10
  MIN_SCORE: 0.1
INDIVIDUAL_SPEC: {SPACE_SIZE: [2000.0, 2000.0, 2000.0], VOXELS_PER_AXIS: [64, 64, 64]}
"""


def test_yaml_overlay_and_unknown_key_rule(tmp_path):
    f = tmp_path / "shelf.yaml"
    f.write_text(SHELF_YAML)
    cfg = CFG.update_config(str(f))
    assert cfg.MODEL == "faster_voxelpose" and cfg.DATASET.NUM_JOINTS == 17 and cfg.TEST.BATCH_SIZE == 16
    assert isinstance(cfg.DATASET.HEATMAP_SIZE, np.ndarray) and list(cfg.CAPTURE_SPEC.VOXELS_PER_AXIS) == [80, 80, 20]
    assert cfg.TRAIN.LAMBDA_LOSS_FUSED == 5.0 and cfg.RESNET.NUM_LAYERS == 50          # untouched defaults
    assert cfg["DATASET"]["CAMERA_NUM"] == cfg.DATASET.CAMERA_NUM == 5
    # a fresh tree every time
    assert CFG.default_config().DATASET.NUM_JOINTS == 15
    bad = tmp_path / "bad.yaml"
    bad.write_text("DATASET:\n  NUM_JOINTZ: 3\n")
    with pytest.raises(ValueError, match="DATASET.NUM_JOINTZ not exist"):
        CFG.update_config(str(bad))
    bad.write_text("NO_SUCH_SECTION: 1\n")
    with pytest.raises(ValueError, match="NO_SUCH_SECTION not exist"):
        CFG.update_config(str(bad))


def test_config_tree_drives_the_model_constructor(tmp_path):
    """The YAML-built tree has every field the hot path reads (same shapes as synthetic.make_cfg)."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    f = tmp_path / "shelf.yaml"
    f.write_text(SHELF_YAML)
    cfg = CFG.update_config(str(f))
    cfg.DEVICE = "cpu"
    m = FV.FasterVoxelPoseNet(cfg, _lib=object())
    ref = FV.FasterVoxelPoseNet(S.make_cfg("shelf", device="cpu"), _lib=object())
    assert list(m.state_dict()) == list(ref.state_dict())


def test_checkpoint_flavours(tmp_path):
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("tiny", device="cpu")
    model = FV.FasterVoxelPoseNet(cfg, _lib=object())
    sd = S.fill_state_dict(model.state_dict(), seed=3)
    best = tmp_path / "model_best.pth.tar"                      # bare state_dict (utils.py:92-98)
    torch.save(sd, best)
    full = tmp_path / "checkpoint.pth.tar"                      # full checkpoint with a backbone and DataParallel prefix
    # as run/train.py writes it: 'precision' is an np.float64 (np.mean of the APs), plus an optimizer state
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=1e-4)
    torch.save({"epoch": 7, "state_dict": {**{"module." + k: v for k, v in sd.items()},
                                            "module.backbone.conv1.weight": torch.zeros(4)},
                "precision": np.float64(0.875), "aps": np.array([0.5, 0.75]), "optimizer": opt.state_dict()}, full)
    for path in (best, full):
        m = FV.FasterVoxelPoseNet(cfg, _lib=object())
        rep = CK.load_model_file(m, str(path))
        assert rep == dict(missing=[], unexpected=[], shape_mismatch=[])
        got = m.state_dict()
        assert all(torch.equal(got[k], sd[k]) for k in sd)
    broken = dict(sd)
    broken.pop(next(iter(broken)))
    torch.save(broken, best)
    with pytest.raises(ValueError, match="1 missing"):
        CK.load_model_file(FV.FasterVoxelPoseNet(cfg, _lib=object()), str(best))
    with pytest.raises(ValueError, match="neither a state_dict"):
        torch.save([1, 2, 3], best)
        CK.read_state_dict(str(best))


def test_affine_transform_matches_reference_golden():
    """get_affine_transform / get_scale against outputs of the reference's own functions
    (tests/golden/make_golden_transforms.py; lib/utils/transforms.py:15,81)."""
    from faster_voxelpose_amd.utils import transforms as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz"))
    for i in range(len(g["rot"])):
        a = T.get_affine_transform(g["center"][i], g["scale"][i], g["rot"][i], g["out"][i], shift=g["shift"][i],
                                   inv=int(g["inv"][i]))
        assert np.array_equal(a, g["affine"][i]), i
    for (a, b, c, d), want in zip(g["sizes"], g["scales"]):
        got = T.get_scale((int(a), int(b)), (int(c), int(d)))
        assert got.dtype == np.float32 and np.array_equal(got, want)
    # scalar scale and torch inputs are accepted like the reference does
    t1 = T.get_affine_transform(torch.tensor([5.0, 6.0]), torch.tensor([2.0, 3.0]), 30, (64, 48))
    t2 = T.get_affine_transform(np.array([5.0, 6.0]), np.array([2.0, 3.0], np.float32), 30, (64, 48))
    assert np.array_equal(t1, t2)
    assert T.get_affine_transform(np.array([5.0, 6.0]), 2.0, 0, (64, 48)).shape == (2, 3)
