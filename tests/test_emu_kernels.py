"""Kernel-logic tests without a GPU: the unmodified HIP sources, compiled for the host
against tests/hipemu, driven through the same C ABI and the same Python host code as the
product, compared with the golden vectors and the oracle.  (The emulator is test
infrastructure; the product never loads it.)"""
import ctypes as C
import os

os.environ.setdefault("FVP_BB_BIG_MIN_TILES", "1")   # emulated backbone runs exercise the large-tile kernel too

import numpy as np
import pytest
import torch

import fvp_oracle as O
from cases import make_inputs, make_weights
from common import check_outputs, front7_stack, load_golden, reg_stack, run_custom_conv_stack, split_k_stack
import fvp_synthetic as S
from faster_voxelpose_amd.models import faster_voxelpose as FV


def build_model(case, lib):
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case)
    model = FV.FasterVoxelPoseNet(cfg, _lib=lib)
    model.load_state_dict(make_weights(case, model.state_dict()))
    return model, cfg, cams, seq, rt, heat, meta


@pytest.mark.parametrize("case", ["tiny_g_b2_all", "tiny_u_b3_thr"])
def test_pipeline_matches_reference_golden(case, emu_lib):
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    g = load_golden(case)
    # default forward: the whole-space cubes are never materialised (fvp_project_columns feeds C2CNet)
    with torch.no_grad():
        fused, planes, centers, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    assert model.engine.last["cubes"] is None
    report = {}
    check_outputs(case, g, fused, planes, centers, model.engine, report)
    print(report)
    feat1d = model.engine.last["feat1d"].clone()
    # materialising forward: cubes bit-equal to the reference, and everything downstream identical to the default
    model.engine.keep_hdn_cubes = True
    with torch.no_grad():
        f2, p2, c2, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    check_outputs(case, g, f2, p2, c2, model.engine)
    assert torch.equal(model.engine.last["feat1d"], feat1d)
    assert torch.equal(fused, f2) and torch.equal(planes, p2) and torch.equal(centers, c2)


def test_materialised_path_equals_fused_path(emu_lib):
    """fvp_project_individual + fvp_triplane_max == fvp_project_individual_triplane, bit for bit,
    and the drop-in ProjectLayer.forward reproduces the reference's cubes."""
    case = "tiny_g_b2_all"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    g = load_golden(case)
    with torch.no_grad():
        model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        planes_fused = model.engine.last_jln["planes"].clone()
        model.joint_net.fused_projection = False
        fused, planes, centers, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        assert torch.equal(model.engine.last_jln["planes"], planes_fused)
        pc = torch.from_numpy(g["proposal_centers_hdn"][0])
        cubes, offset = model.joint_net.project_layer(heat, 0, meta, pc, cams, rt)
    assert np.array_equal(cubes.numpy(), g["jl0_cubes"])
    assert np.array_equal(offset.numpy(), g["jl0_offset"])


def test_sample_grid_and_whole_projection_drop_in(emu_lib):
    case = "tiny_u_b3_thr"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    g = load_golden(case)
    pl = model.pose_net.project_layer
    with torch.no_grad():
        cubes = pl(heat, meta, cams, rt)
    stride = int(g["grid_stride"])
    assert np.array_equal(pl.sample_grid[seq][:, 0, ::stride].numpy(), g["grid_digest"]), "sampling grid not bit-equal"
    sx, sy = g["cubes_sub_stride"]
    assert np.array_equal(cubes[:, :, ::sx, ::sy, :].numpy(), g["cubes_sub"])


def test_nms_topk_edge_cases(emu_lib):
    """Plateaus, -inf padding at the border, negative maxima below the zero background,
    tie -> lowest flat index; same as the oracle's rule."""
    from faster_voxelpose_amd.core.proposal import nms2D
    rng = np.random.default_rng(0)
    m = torch.from_numpy(rng.normal(size=(3, 1, 12, 12)).astype(np.float32))
    m[0, 0, 0, 0] = 5.0
    m[0, 0, 11, 11] = 5.0            # tie between first and last cell
    m[1, 0, 4, 4] = m[1, 0, 4, 5] = 7.0   # plateau
    m[2] = -1.0                      # everything negative: all cells are maxima of a constant map
    vals, idx, flat = nms2D(m, 6, _lib=emu_lib)
    ov, oi, of = O.nms2d(m, 6)
    assert torch.equal(flat, of) and torch.equal(idx, oi) and torch.equal(vals, ov)
    assert flat[0, 0].item() == 0 and flat[0, 1].item() == 143
    assert flat[2].tolist() == [0, 1, 2, 3, 4, 5]
    # more than 16 384 cells: keep bits + in-place LDS map instead of register-resident cells
    big = torch.from_numpy(rng.normal(size=(2, 1, 136, 128)).astype(np.float32))
    big[0, 0, 0, 0] = big[0, 0, 135, 127] = 9.0
    big[1, 0, 70, 3] = big[1, 0, 70, 4] = 8.0
    vals, idx, flat = nms2D(big, 5, _lib=emu_lib)
    ov, oi, of = O.nms2d(big, 5)
    assert torch.equal(flat, of) and torch.equal(idx, oi) and torch.equal(vals, ov)


def test_conv_stack_matches_oracle_on_ragged_batch(emu_lib):
    """P2PNet on 5 planes (not a multiple of the tile) with two planes masked out."""
    case = "tiny_g_b2_all"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    J, Cn = cfg.DATASET.NUM_JOINTS, cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS[0]
    x = torch.from_numpy(np.random.default_rng(1).random((5, J, Cn, Cn), dtype=np.float32))
    sd = {k: v for k, v in model.state_dict().items()}
    want = O.p2p_net(sd, "joint_net.conv_net", x)
    got = model.joint_net.conv_net(x)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=3e-6, atol=3e-6)
    valid = torch.tensor([1, 0, 1, 1, 0], dtype=torch.uint8)
    got2 = model.joint_net.conv_net(x, plane_valid=valid, valid_div=1)
    np.testing.assert_allclose(got2[[0, 2, 3]].numpy(), want[[0, 2, 3]].numpy(), rtol=3e-6, atol=3e-6)
    # 1-D stack
    z = torch.from_numpy(np.random.default_rng(2).random((7, J, cfg.CAPTURE_SPEC.VOXELS_PER_AXIS[2]), dtype=np.float32))
    np.testing.assert_allclose(model.pose_net.c2c_net(z).numpy(), O.c2c_net(sd, "pose_net.c2c_net", z).numpy(),
                               rtol=3e-6, atol=3e-6)


def test_conv_ragged_masked_batch_and_centernet(emu_lib):
    """Ragged plane count with masked planes through P2PNet, and CenterNet on its own."""
    case = "tiny_g_b2_all"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    J, Cn = cfg.DATASET.NUM_JOINTS, cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS[0]
    x = torch.from_numpy(np.random.default_rng(5).random((11, J, Cn, Cn), dtype=np.float32))
    sd = dict(model.state_dict())
    want = O.p2p_net(sd, "joint_net.conv_net", x)
    valid = torch.tensor([1, 0, 0, 1, 1, 0, 1, 1, 1, 0, 1], dtype=torch.uint8)
    got = model.joint_net.conv_net(x, plane_valid=valid, valid_div=1)
    keep = valid.bool()
    np.testing.assert_allclose(got[keep].numpy(), want[keep].numpy(), rtol=3e-6, atol=3e-6)
    got_all = model.joint_net.conv_net(x)
    np.testing.assert_allclose(got_all.numpy(), want.numpy(), rtol=3e-6, atol=3e-6)
    # CenterNet geometry (16x16 map of the tiny config) through the same path
    X, Y, Z = cfg.CAPTURE_SPEC.VOXELS_PER_AXIS
    cubes = torch.from_numpy(np.random.default_rng(6).random((3, J, X, Y, Z), dtype=np.float32))
    hm_w, sz_w = O.center_net(sd, "pose_net.center_net", cubes)
    hm, sz = model.pose_net.center_net(cubes)
    np.testing.assert_allclose(hm.numpy(), hm_w.numpy(), rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(sz.numpy(), sz_w.numpy(), rtol=2e-5, atol=5e-5)


def test_fused_c2c_equals_generic_interpreter(emu_lib):
    """fvp_conv_stack_run_fused_1d (whole C2CNet in one kernel, K split over four wave groups) vs the per-op
    interpreter (one k-ordered chain): equal up to the rounding of the four-way partial sums, and both match
    the oracle."""
    case = "tiny_g_b2_all"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    J, Z = cfg.DATASET.NUM_JOINTS, cfg.CAPTURE_SPEC.VOXELS_PER_AXIS[2]
    z = torch.from_numpy(np.random.default_rng(8).random((7, J, Z), dtype=np.float32))
    sd = dict(model.state_dict())
    model.engine.fused_c2c = True
    fused = model.pose_net.c2c_net(z)
    model.engine.fused_c2c = False
    generic = model.pose_net.c2c_net(z)
    np.testing.assert_allclose(fused.numpy(), generic.numpy(), rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(fused.numpy(), O.c2c_net(sd, "pose_net.c2c_net", z).numpy(), rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(generic.numpy(), O.c2c_net(sd, "pose_net.c2c_net", z).numpy(), rtol=3e-6, atol=3e-6)


def test_rasteriser_kernel_matches_reference_golden(emu_lib):
    """fvp_rasterise_heatmaps (emulated) vs the reference's generate_input_heatmap outputs: exact."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from heatmap_cases import HEATMAP_CASES, make_pred2d
    from faster_voxelpose_amd.dataset import generate_input_heatmaps
    for case in HEATMAP_CASES:
        cfg, all_preds, rt, sigma = make_pred2d(case)
        g = load_golden(case)["heatmaps"]
        hm, cl = generate_input_heatmaps(all_preds, rt, cfg, sigma=sigma, device="cpu", channels_last=True, _lib=emu_lib)
        assert np.array_equal(hm.numpy(), g), case
        J = cfg.DATASET.NUM_JOINTS
        V, _, H, W = g.shape
        assert np.array_equal(cl.numpy()[..., :J], g.reshape(V, J, H * W).transpose(0, 2, 1))
        assert not cl.numpy()[..., J:].any()


def test_backbone_emulated_matches_bf16_oracle(emu_lib):
    """The whole Pose-ResNet-50 through fvp_bb_run (emulated bf16 MFMA) on a 32x32 image: as close to
    the fp32 evaluation as the oracle's bf16-emulating mode is (two bf16 evaluations with different
    summation orders agree with each other only to the same ~1 %), both heatmap layouts identical."""
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    cfg = CFG.default_config()
    cfg.DEVICE = "cpu"
    m = RN.PoseResNet(cfg, _lib=emu_lib)
    sd = S.fill_backbone_state_dict(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    x = torch.from_numpy(np.random.default_rng(0).random((1, 3, 32, 32), dtype=np.float32))
    y = m(x)
    o32, o16 = O.pose_resnet(sd, x), O.pose_resnet(sd, x, bf16=True)
    e_prod = float((y - o32).norm() / o32.norm())
    e_orc = float((o16 - o32).norm() / o32.norm())
    assert e_prod < 1.5 * e_orc + 1e-3, (e_prod, e_orc)
    cl = m.forward_channels_last(x)
    J = cfg.DATASET.NUM_JOINTS
    assert torch.equal(cl[..., :J], y.reshape(1, J, -1).permute(0, 2, 1)) and not cl[..., J:].any()


@pytest.mark.parametrize("hw", [(32, 32), (64, 160)])
def test_backbone_fused_stem_pool_equals_the_two_launches(emu_lib, monkeypatch, hw):
    """k_bb_stem_pool (conv1 + bn1 + ReLU + max-pool in one kernel, round 6) and k_bb_bottleneck64 (a layer1 bottleneck in one
    kernel, with and without the downsample branch) against the layer-by-layer launches (FVP_BB_NO_FUSE_STEM / _BLOCK,
    diagnostics build): the same MFMA chains and roundings per output, so the stage's output tensor must be bit-equal - on an
    image smaller than one tile and on one with ragged tiles and two tile columns.  Only the stem and layer1 are run (the op
    list up to the last layer1 block through fvp_bb_run): the rest of the network is covered by the test above."""
    import ctypes as C
    from faster_voxelpose_amd import _capi as capi
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    cfg = CFG.default_config()
    cfg.DEVICE = "cpu"
    m = RN.PoseResNet(cfg, _lib=emu_lib)
    m.load_state_dict(S.fill_backbone_state_dict(m.state_dict(), seed=5))
    N, (H, W) = 2, hw
    x = torch.from_numpy(np.random.default_rng(1).random((N, 3, H, W), dtype=np.float32))
    plan = m._plan(H, W)
    m.ensure_packed(plan)
    nops = 1 + max(i for i, o in enumerate(m._convs) if str(o.get("key", "")).startswith("layer1."))
    last = plan["ops"][nops - 1].dst

    def run():
        bufs = []
        for name in plan["names"]:
            c, h, w = plan["shapes"][name]
            bufs.append(torch.zeros((N, h, w // 2, c) if name == "x" else (N, h, w, c), dtype=torch.bfloat16))
        capi.check(emu_lib, emu_lib.fvp_bb_input(C.c_void_p(x.data_ptr()), C.c_void_p(bufs[0].data_ptr()), N, 3, H, W, None), "input")
        arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
        capi.check(emu_lib, emu_lib.fvp_bb_run(plan["ops"], nops, C.c_void_p(m._wblob.data_ptr()), C.c_void_p(m._eblob.data_ptr()),
                                               arr, len(bufs), N, None, 0, None, None), "fvp_bb_run")
        return bufs[last].clone()

    fused = run()
    monkeypatch.setenv("FVP_BB_NO_FUSE_STEM", "1")
    no_stem = run()
    monkeypatch.setenv("FVP_BB_NO_FUSE_BLOCK", "1")
    plain = run()
    assert float(fused.float().abs().max()) > 0
    assert torch.equal(fused.view(torch.int16), no_stem.view(torch.int16)), "stem"
    assert torch.equal(fused.view(torch.int16), plain.view(torch.int16)), "bottleneck"


def test_cached_fine_grid_gives_identical_planes(emu_lib):
    """The fused tri-plane kernel with the per-sequence coordinate cache (fine_grid) == recomputed projection."""
    case = "tiny_g_b2_all"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    with torch.no_grad():
        f1, p1, c1, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        planes1 = model.engine.last_jln["planes"].clone()
        model.engine.cache_fine_grid = True
        f2, p2, c2, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    assert model.engine.geo.fine_grid is not None
    assert torch.equal(model.engine.last_jln["planes"], planes1) and torch.equal(f1, f2) and torch.equal(p1, p2)


# ---- edge cases shared with the GPU suite (tests/edge_cases.py) ------------------------------------
import edge_cases as E  # noqa: E402


def test_zero_batch_through_every_export(emu_lib):
    E.zero_batch_through_every_export(emu_lib, "cpu")


def test_negative_bbox_gives_an_empty_window(emu_lib):
    E.negative_bbox_gives_an_empty_window(emu_lib, "cpu")


def test_sampling_grids_equal_reference_campus(emu_lib):
    E.sampling_grids_equal_reference(emu_lib, "cpu", "campus")


def test_winograd_on_maps_that_do_not_divide_the_workgroup_tile(emu_lib, monkeypatch):
    """CenterNet on a 20 x 12 detection grid with FVP_WINO_GENERIC=1: 6 tiles per row do not divide the workgroup's
    128 tiles (masked lanes, two planes per unit) -- against the oracle's plain fp32 conv stack.  (The library reads
    the switch once at load time: the session's emulated library is loaded by conftest with it set.)"""
    monkeypatch.setenv("FVP_WINO_GENERIC", "1")
    cfg = S.make_cfg("tiny", device="cpu", voxels=[20, 12, 8])
    model = FV.FasterVoxelPoseNet(cfg, _lib=emu_lib)
    sd = S.fill_state_dict(model.state_dict(), seed=5)
    model.load_state_dict(sd)
    assert any(o.wino_off > 0 for o in model.engine.specs["center_net"].op_array), "fixture must exercise the Winograd path"
    rng = np.random.default_rng(2)
    cubes = torch.from_numpy(rng.random((3, cfg.DATASET.NUM_JOINTS, 20, 12, 8), dtype=np.float32))
    hm_w, sz_w = O.center_net(sd, "pose_net.center_net", cubes)
    with torch.no_grad():
        hm, sz = model.pose_net.center_net(cubes)
    np.testing.assert_allclose(hm.numpy(), hm_w.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sz.numpy(), sz_w.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("cin,cmid,hw,planes", [(64, 128, (16, 20), 3), (128, 64, (24, 24), 2)])
def test_split_k_direct_conv_on_small_maps(emu_lib, cin, cmid, hw, planes):
    """k_conv_dma<..., KS>: 3x3 layers with >= 64 channels on small non-power-of-two maps split the reduction over the
    four waves of a workgroup (CenterNet's 20x20 / 40x40 levels).  Against a float64 torch evaluation, ragged plane
    count, residual + ReLU epilogue; maps of 256 .. 576 pixels take the split form (rows of 20 and 24: masked lanes)."""
    spec, w, ref, o = split_k_stack(cin, cmid, hw, seed=cin)
    x = torch.from_numpy(np.random.default_rng(5).normal(size=(planes, cin) + hw).astype(np.float32))
    got = run_custom_conv_stack(emu_lib, "cpu", spec, w, x)[o]
    want = ref(x)
    np.testing.assert_allclose(got.double().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("cin,hw,planes", [(15, (10, 64), 3), (17, (8, 64), 2), (3, (4, 128), 1), (15, (8, 80), 2)])
def test_front_conv7_on_16x16x4_tiles(emu_lib, cin, hw, planes):
    """k_conv7: the 7x7 front conv with the reduction ordered (channel group of 4, kernel row, kernel column) - widths 64 /
    128, 15 and 17 joints (4 and 5 channel groups; 15 and 17 are not multiples of 4: zero channels), 3 channels (three
    all-zero groups), a height that is not a multiple of the 4-row tile, a masked plane; the 80-wide map stays on the
    pixel-pair form of k_conv_dma.  Against a float64 torch evaluation."""
    spec, w, ref, o = front7_stack(cin, hw, seed=cin)
    x = torch.from_numpy(np.random.default_rng(7).normal(size=(planes, cin) + hw).astype(np.float32))
    got = run_custom_conv_stack(emu_lib, "cpu", spec, w, x)[o]
    want = ref(x)
    np.testing.assert_allclose(got.double().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
    if planes > 1:
        valid = torch.ones(planes, dtype=torch.uint8)
        valid[1] = 0
        got2 = run_custom_conv_stack(emu_lib, "cpu", spec, w, x, plane_valid=valid)[o]
        keep = valid.bool()
        assert torch.equal(got2[keep], got[keep])


@pytest.mark.parametrize("fused_head,head_cout", [(True, 15), (False, 15), (True, 17)])
def test_register_direct_conv_equals_the_staged_kernel(emu_lib, monkeypatch, fused_head, head_cout):
    """k_conv_reg (1x1 convs, transposed convs, transposed conv + fused 1x1 head: activations straight from the map into
    MFMA operands, weights resident in LDS) against k_conv_dma on the same stack: the same bits (same ascending-channel
    MFMA chain), and both against a float64 torch evaluation.  Five planes: the last workgroup's waves run out of tiles."""
    spec, w, ref, outs = reg_stack(seed=3, fused_head=fused_head, head_cout=head_cout)   # 17: Shelf / Campus joints
    x = torch.from_numpy(np.random.default_rng(9).normal(size=(5, 32, 16, 16)).astype(np.float32))
    monkeypatch.setenv("FVP_CONV_REG_MIN_TILES", "1")
    got = run_custom_conv_stack(emu_lib, "cpu", spec, w, x)
    valid = torch.tensor([1, 0, 1, 1, 0], dtype=torch.uint8)            # masked planes: their tiles are skipped
    masked = run_custom_conv_stack(emu_lib, "cpu", spec, w, x, plane_valid=valid)
    monkeypatch.setenv("FVP_CONV_REG_MIN_TILES", "1000000000")
    staged = run_custom_conv_stack(emu_lib, "cpu", spec, w, x)
    want = ref(x)
    for name, o in outs.items():
        assert torch.equal(got[o], staged[o]), name
        assert torch.equal(masked[o][valid.bool()], got[o][valid.bool()]), name
        np.testing.assert_allclose(got[o].double().numpy(), want[name].numpy(), rtol=2e-5, atol=2e-5, err_msg=name)


@pytest.mark.parametrize("quad", [False, True])
@pytest.mark.parametrize("cap", [12, 40, 100])
def test_fused_projection_one_tile_two_tile_and_gather_rectangles(emu_lib, monkeypatch, cap, quad):
    """The LDS-staged fused projection treats a (block, view) rectangle in one of three ways: it fits one tile (DMA one
    view ahead), it fits the two tiles together (staged when its turn comes, round 3), or it is gathered from global
    memory.  FVP_TRI_CAP_PX shrinks the size the kernel regards as fitting so that all three occur on the miniature
    fixture; the planes must stay bit-equal to the materialised path whatever the mix - for the lane-per-voxel form
    and for the four-lanes-per-voxel form (FVP_TRIPLANE_QUAD)."""
    case = "tiny_g_b2_all"
    model, cfg, cams, seq, rt, heat, meta = build_model(case, emu_lib)
    with torch.no_grad():
        model.joint_net.fused_projection = False
        model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        want = model.engine.last_jln["planes"].clone()
        model.joint_net.fused_projection = True
        monkeypatch.setenv("FVP_TRI_CAP_PX", str(cap))
        monkeypatch.setenv("FVP_TRI_TWO_TILE", "1")
        # (the staged forms are diagnostics-build code since round 6: the emulator library is such a build)
        monkeypatch.setenv("FVP_TRIPLANE_QUAD" if quad else "FVP_TRIPLANE_LANE", "1")
        model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        got = model.engine.last_jln["planes"].clone()
    assert want.abs().sum() > 0 and torch.equal(got, want)



@pytest.mark.parametrize("J,V", [(1, 3), (32, 3), (21, 3), (15, 3), (17, 3), (19, 3), (5, 1), (5, 8)])
def test_joint_and_view_count_extremes(emu_lib, J, V):
    """The limits include/fvp.h states (FVP_MAX_JOINTS = 32, FVP_MAX_VIEWS = 8) and the minima (one joint, one view; J = 21
    -> JP = 24, a channel padding none of the shipped configs has): the whole forward on the miniature shape against the
    oracle - proposal centres / valid flags exact, joints inside the random-weight noise floor."""
    import copy
    cfg = S.make_cfg("tiny", device="cpu", min_score=-1.0)
    cfg.DATASET.NUM_JOINTS, cfg.DATASET.CAMERA_NUM = J, V
    cams0, seq = S.load_cameras("tiny")
    base = cams0[seq]
    cl = []
    for i in range(V):                                  # cameras beyond the three of the fixture: shifted copies
        c = copy.deepcopy(base[i % len(base)])
        c["T"] = (np.asarray(c["T"], np.float64).reshape(3, 1) + np.array([[37.0 * i], [-21.0 * i], [5.0 * i]])).tolist()
        cl.append(c)
    cams = {seq: cl}
    rt = S.resize_transform(cfg)
    heat = S.heatmaps_uniform(cfg, 2, seed=4)
    model = FV.FasterVoxelPoseNet(cfg, _lib=emu_lib)
    sd = S.fill_state_dict(model.state_dict(), seed=3)
    model.load_state_dict(sd)
    meta = {"seq": [seq] * 2}
    with torch.no_grad():
        fused, planes, centers, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    of, op, oc = O.Oracle(cfg, sd).forward(heat, meta, cams, rt)
    assert fused.shape == (2, cfg.CAPTURE_SPEC.MAX_PEOPLE, J, 5) and planes.shape == (3, 2, cfg.CAPTURE_SPEC.MAX_PEOPLE, J, 2)
    assert torch.equal(centers[..., :4], oc[..., :4])
    np.testing.assert_allclose(fused[..., :3].numpy(), of[..., :3].numpy(), rtol=0, atol=2e-2)     # mm
    np.testing.assert_allclose(fused[..., 4].numpy(), of[..., 4].numpy(), rtol=2e-4, atol=1e-6)
    # the fused projection (J = 15 -> JP = 16: the unstaged compact-block form the Panoptic shape takes) against the
    # materialised cubes + tri-plane maxima: bit for bit
    planes_fused = model.engine.last_jln["planes"].clone()
    model.joint_net.fused_projection = False
    with torch.no_grad():
        model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    assert torch.equal(model.engine.last_jln["planes"], planes_fused)


@pytest.mark.parametrize("net,args", [("centernet", (15, 80, 80)), ("centernet", (17, 80, 80)), ("p2pnet", (15, 15, 64)),
                                      ("p2pnet", (17, 17, 64)), ("p2pnet", (15, 15, 128)), ("c2cnet", (15, 20))])
def test_packed_parameter_blob_layout(emu_lib, net, args):
    """netspec.finalize() (host) and fvp_pack_conv (library) must agree on the parameter blob: every op's copies - packed
    weights, epilogue vectors, the Winograd-domain copy, the pixel-pair + k-grouped copies of the 7x7 front conv, the
    column-tap pairs of the transposed convs - at the BASELINE shapes.  Each op is packed into a blob of sentinels: what it
    writes lies inside [64, nparams), is disjoint from every other op's writes, and the first 64 floats (the zero page)
    and the guard behind the blob stay untouched."""
    import ctypes as C

    from faster_voxelpose_amd import _capi as capi
    from faster_voxelpose_amd import netspec
    spec = {"centernet": netspec.centernet_spec, "p2pnet": netspec.p2pnet_spec, "c2cnet": netspec.c2cnet_spec}[net](*args)
    guard = 8192
    sentinel = 12345.678
    blob = torch.full((spec.nparams + guard,), sentinel)
    g = torch.Generator().manual_seed(3)
    owner = torch.full((spec.nparams + guard,), -1, dtype=torch.int32)
    for key, bn, transposed, oi in spec.param_keys:
        shape = spec.entries[key + ".weight"][0]
        w = (torch.rand(shape, generator=g) + 0.5).contiguous()       # never equal to the sentinel, never zero
        b = torch.rand(shape[1] if transposed else shape[0], generator=g) + 0.5
        bnp = [None] * 4
        if bn is not None:
            c = spec.entries[bn + ".weight"][0]
            bnp = [torch.rand(c, generator=g) + 0.5 for _ in range(4)]
        before = blob.clone()
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        capi.check(emu_lib, emu_lib.fvp_pack_conv(ptr(w), ptr(b), *[ptr(t) for t in bnp], 1e-5, 1 if transposed else 0,
                                                  C.byref(spec.op_array[oi]), ptr(blob), None), "fvp_pack_conv")
        touched = (blob != before).nonzero().flatten()
        assert touched.numel() > 0, key
        assert int(touched.min()) >= 64 and int(touched.max()) < spec.nparams, (key, int(touched.min()), int(touched.max()))
        assert bool((owner[touched] == -1).all()), "%s writes into the region of op %d" % (key, int(owner[touched].max()))
        owner[touched] = oi
    assert bool((blob[:64] == sentinel).all()) and bool((blob[spec.nparams:] == sentinel).all())
    # ... and nothing inside is left unwritten (a kernel would read it) beyond the round-up-to-4 gaps between regions
    unwritten = int((owner[64:spec.nparams] == -1).sum())
    assert unwritten <= 3 * 2 * len(spec.param_keys), unwritten


def test_front_conv7_random_shapes(emu_lib):
    """k_conv7 over random (joints, height, planes, mask) draws: heights 4..19 (every remainder of the 4-row tile), 1..20
    channels (1..5 channel groups, ragged last group), masked planes.  Against a float64 torch evaluation."""
    rng = np.random.default_rng(11)
    for _ in range(8):
        cin = int(rng.integers(1, 21))
        h = int(rng.integers(4, 20))
        w_ = int(rng.choice([64, 128]))
        planes = int(rng.integers(1, 4))
        spec, w, ref, o = front7_stack(cin, (h, w_), seed=cin + h)
        x = torch.from_numpy(rng.normal(size=(planes, cin, h, w_)).astype(np.float32))
        valid = torch.from_numpy((rng.random(planes) < 0.7).astype(np.uint8))
        valid[0] = 1
        got = run_custom_conv_stack(emu_lib, "cpu", spec, w, x, plane_valid=valid)[o]
        keep = valid.bool()
        np.testing.assert_allclose(got[keep].double().numpy(), ref(x[keep]).numpy(), rtol=2e-5, atol=2e-5,
                                   err_msg=str((cin, h, w_, planes)))


def test_validate_is_a_drop_in_for_the_reference_loop(emu_lib, tmp_path):
    """core.function.validate with the reference's signature (lib/core/function.py:117: config, backbone, model, loader,
    output_dir, has_evaluate_function) and loader protocol - 4-tuples (inputs, targets, meta, input_heatmaps), the dataset
    carrying cameras / resize_transform / evaluate - exactly what run/validate.py:97-102 hands over.  Stub loader, the
    real (emulated) hot path; the poses handed to dataset.evaluate equal the plain forward's, bit for bit."""
    from faster_voxelpose_amd.core import function as FN
    model, cfg, cams, seq, rt, heat, meta = build_model("tiny_g_b2_all", emu_lib)
    with torch.no_grad():
        want, _, _, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    seen = {}

    class Dataset:
        cameras = cams
        resize_transform = rt.numpy()

        def evaluate(self, all_fused_poses):
            seen["poses"] = all_fused_poses.clone()
            return 12.5, "stub metric message"

    class Loader:
        dataset = Dataset()

        def __len__(self):
            return 2

        def __iter__(self):
            for _ in range(2):
                yield torch.zeros(heat.shape[0], heat.shape[1], 3, 8, 8), None, meta, heat

    cfg.DATASET.TEST_HEATMAP_SRC = "pred"                 # heatmaps come from the loader (function.py:142-148)
    import types
    cfg.TEST = types.SimpleNamespace(VISUALIZATION=False)
    cfg.PRINT_FREQ = 1
    assert FN.validate(cfg, None, model, Loader(), str(tmp_path), has_evaluate_function=False) == 0.0
    metric = FN.validate(cfg, None, model, Loader(), str(tmp_path), has_evaluate_function=True)
    assert metric == 12.5
    assert seen["poses"].shape[0] == 2 * want.shape[0]
    assert torch.equal(seen["poses"][:want.shape[0]], want) and torch.equal(seen["poses"][want.shape[0]:], want)
