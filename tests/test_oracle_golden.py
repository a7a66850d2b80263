"""Pin the CPU oracle against the golden vectors captured from the reference
(tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

import fvp_oracle as O
from cases import CASES, make_inputs, make_weights
from common import FLOOR_RULE, floor_rule_check, load_golden
import fvp_synthetic as S


def state_dict_for(cfg, wseed):
    return S.fill_state_dict(O.reference_state_dict_shapes(cfg), seed=wseed)


def test_checkpoint_layout_restatement_matches_product_modules():
    """The oracle's independent restatement of the reference's state_dict layout and the
    product's module tree agree key for key (order and shapes) with each other and with the
    reference's own 485-entry key list (tests/golden/grids.npz, make_golden_grids.py)."""
    import os
    from common import GOLDEN_DIR
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    ref = np.load(os.path.join(GOLDEN_DIR, "grids.npz"))
    for name in ("panoptic", "shelf", "campus", "tiny"):
        cfg = S.make_cfg(name, device="cpu")
        want = O.reference_state_dict_shapes(cfg)
        got = FV.FasterVoxelPoseNet(cfg, _lib=object()).state_dict()
        assert list(want) == list(got)
        assert all(want[k].shape == got[k].shape for k in want)
        if name != "tiny":
            assert list(got) == [str(k) for k in ref[f"{name}_keys"]] and len(got) == 485
            assert [",".join(str(int(d)) for d in v.shape) for v in got.values()] == [str(v) for v in ref[f"{name}_shapes"]]


def test_oracle_fine_grid_matches_reference_digest():
    """The oracle's projection of the joint stage's fine grid (the reference caches it per sequence,
    project_individual.py:82-94) vs the reference's digest: bit-equal."""
    import os
    from common import GOLDEN_DIR
    ref = np.load(os.path.join(GOLDEN_DIR, "grids.npz"))
    sx, sy, sz = (int(v) for v in ref["fine_stride"])
    for name in ("panoptic", "campus"):
        cfg = S.make_cfg(name, device="cpu")
        cams, seq = S.load_cameras(name)
        rt = S.resize_transform(cfg)
        spec = O.IndividualSpec(cfg)
        fine = [int(v) for v in spec.fine]
        assert fine == list(ref[f"{name}_fine_dims"])
        pts = spec.fine_points(torch.zeros(3, dtype=torch.int64), torch.tensor(fine))
        for v in range(len(cams[seq])):
            grid = O.sample_grid(pts, cams[seq][v], cfg, rt).view(*fine, 2)
            assert np.array_equal(grid[::sx, ::sy, ::sz].numpy(), ref[f"{name}_fine"][v]), (name, v)


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_golden(case):
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case)
    g = load_golden(case)
    orc = O.Oracle(cfg, make_weights(case, O.reference_state_dict_shapes(cfg)))
    fused, planes, centers = orc.forward(heat, meta, cams, rt)
    # sampling grid: bit-equal (same torch ops in the same order)
    stride = int(g["grid_stride"])
    assert np.array_equal(orc._grids[seq][:, ::stride].numpy(), g["grid_digest"])
    # cubes: own bilinear restatement vs F.grid_sample: <= 2 ulp of 1.0
    sx, sy = g["cubes_sub_stride"]
    np.testing.assert_allclose(orc.trace["cubes"][:, :, ::sx, ::sy, :].numpy(), g["cubes_sub"], rtol=0, atol=2.5e-7)
    np.testing.assert_allclose(orc.trace["hm2d"][:, 0].numpy(), g["hm2d"], rtol=0, atol=2e-5)
    # integer / index work: exact
    ti = orc.trace["topk_index"].numpy()
    X = cfg.CAPTURE_SPEC.VOXELS_PER_AXIS[0]
    assert np.array_equal(ti[..., 0] * X + ti[..., 1], g["topk_flat"])
    c = centers.numpy()
    assert np.array_equal(c[..., :3], g["proposal_centers"][..., :3])
    assert np.array_equal(c[..., 3], g["proposal_centers"][..., 3])
    v = g["valid"]
    for f in range(v.shape[0]):
        if f"jl{f}_tl" not in g:
            continue
        tl, start, end = orc.trace["jln"][f]["boxes"]
        assert np.array_equal(tl.numpy(), g[f"jl{f}_tl"])
        assert np.array_equal(start.numpy(), g[f"jl{f}_start"])
        assert np.array_equal(end.numpy(), g[f"jl{f}_end"])
        assert np.array_equal(orc.trace["jln"][f]["offset"].numpy(), g[f"jl{f}_offset"])
    # joints: the oracle uses the reference's own conv kernels, so it sits well inside the floor
    d = np.linalg.norm((fused[..., :3].numpy() - g["fused_poses"][..., :3])[v], axis=-1)
    floor = float(g["margins"][5])
    if case in FLOOR_RULE:
        # Campus: even this restatement on the reference's own conv kernels (another thread count / primitive choice)
        # lands 1.2e-3 mm = 2.5 ulp from the reference's fp32 output - the bar there is common.FLOOR_RULE
        floor_rule_check(case, fused[..., :3].numpy(), g)
    else:
        assert d.max() <= (1e-3 if CASES[case][1] == "c" else max(1e-3, 3 * floor)), (d.max(), floor)
    assert np.all(fused[..., :3].numpy()[~v] == 0)


def test_resize_transform_constants():
    """the 2x3 resize transforms of the three shipped configs."""
    expect = {"panoptic": [[0.474074, 0, 24.888889], [0, 0.474074, 0]],
              "shelf": [[0.775194, 0, 0], [0, 0.775194, 3.224806]],
              "campus": [[2.222222, 0, 0], [0, 2.222222, 0]]}
    for name, e in expect.items():
        rt = S.resize_transform(S.make_cfg(name)).numpy()
        np.testing.assert_allclose(rt, np.array(e), atol=1e-5)


def test_nms_tie_rule_and_padding():
    """max-pool NMS keeps plateaus and uses -inf padding; ties resolve to the lowest index."""
    m = torch.zeros(1, 1, 6, 6)
    m[0, 0, 0, 0] = 1.0
    m[0, 0, 3, 3] = 1.0
    m[0, 0, 3, 4] = 1.0          # plateau: both kept
    m[0, 0, 5, 5] = -2.0         # below the zero background: suppressed to (-)0
    vals, idx, flat = O.nms2d(m, 4)
    assert flat[0].tolist()[:3] == [0, 21, 22]
    assert vals[0].tolist()[:3] == [1.0, 1.0, 1.0]
    assert idx[0, 1].tolist() == [3, 3]


# ---- "next" row f-2: input heatmaps rasterised from 2-D detections ---------------------------------
from heatmap_cases import HEATMAP_CASES, make_pred2d  # noqa: E402


@pytest.mark.parametrize("case", list(HEATMAP_CASES))
def test_oracle_rasteriser_matches_reference_golden(case):
    """The oracle's restatement of JointsDataset.generate_input_heatmap (+ affine_transform of the
    detections) reproduces the reference's heatmaps bit for bit."""
    cfg, all_preds, rt, sigma = make_pred2d(case)
    g = load_golden(case)
    views = [p if len(p) else None for p in all_preds]
    out = []
    for preds in views:
        if preds is None:                       # a view without detections: the reference would index joints[0]
            out.append(torch.zeros(cfg.DATASET.NUM_JOINTS, cfg.DATASET.HEATMAP_SIZE[1], cfg.DATASET.HEATMAP_SIZE[0]))
        else:
            out.append(O.input_heatmaps_from_pred2d([preds], rt, cfg.DATASET.IMAGE_SIZE, cfg.DATASET.HEATMAP_SIZE, sigma)[0])
    hm = torch.stack(out).numpy()
    assert hm.dtype == np.float32 and hm.shape == g["heatmaps"].shape
    assert np.array_equal(hm, g["heatmaps"])


# ---- "next" row f-1: Pose-ResNet-50 backbone ----------------------------------------------------------
def test_oracle_backbone_matches_reference_golden():
    """The functional restatement of resnet.py (BatchNorm folded to scale / shift) reproduces the
    reference's heatmaps; its bf16-emulating mode stays within ~1 % of them; the product module has
    the reference's state_dict keys."""
    from make_golden_backbone import WSEED, inputs
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    g = load_golden("backbone_r50")
    cfg = CFG.default_config()
    cfg.DEVICE = "cpu"
    m = RN.PoseResNet(cfg, _lib=object())
    assert list(m.state_dict()) == list(g["keys"])
    sd = S.fill_backbone_state_dict(m.state_dict(), seed=WSEED)
    x = inputs()
    y32 = O.pose_resnet(sd, x).numpy()
    ref = g["heatmaps"]
    np.testing.assert_allclose(y32, ref, rtol=0, atol=2e-4 * float(np.abs(ref).max()))
    y16 = O.pose_resnet(sd, x, bf16=True).numpy()
    assert np.linalg.norm(y16 - ref) / np.linalg.norm(ref) < 3e-2
