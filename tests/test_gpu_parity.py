"""Parity of the HIP path on a real MI355X: golden vectors from the reference, the CPU oracle,
and size-independent properties at BASELINE.json's full sizes.  Everything goes through the
C ABI (libfvp_hip.so) via the reference-shaped modules."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import fvp_oracle as O
from cases import CASES, make_inputs, make_weights
from common import check_outputs, drift_check, front7_stack, load_golden, reg_stack, run_custom_conv_stack, split_k_stack
import fvp_synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def build(case_or_cfg, wseed=7, case=None):
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = case_or_cfg
    model = FV.get(cfg).to(DEV)
    sd = make_weights(case, model.state_dict()) if case else S.fill_state_dict(model.state_dict(), seed=wseed)
    model.load_state_dict(sd)
    return model, sd


@pytest.mark.parametrize("case", list(CASES))
def test_golden_case(case):
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case, device=DEV)
    model, _ = build(cfg, wseed, case)
    with torch.no_grad():
        fused, planes, centers, hm, loss = model(meta=meta, input_heatmaps=heat.to(DEV), cameras=cams,
                                                 resize_transform=rt.to(DEV))
    torch.cuda.synchronize()
    assert loss is None and hm.shape == heat.shape
    report = {}
    try:
        g = load_golden(case)
        assert model.engine.last["cubes"] is None          # default forward: no [B,J,X,Y,Z] cubes (fvp_project_columns)
        check_outputs(case, g, fused, planes, centers, model.engine, report)
        feat1d = model.engine.last["feat1d"].clone()
        model.engine.keep_hdn_cubes = True                 # materialising forward: cubes checked, same bits downstream
        with torch.no_grad():
            f2, p2, c2, _, _ = model(meta=meta, input_heatmaps=heat.to(DEV), cameras=cams, resize_transform=rt.to(DEV))
        torch.cuda.synchronize()
        check_outputs(case, g, f2, p2, c2, model.engine)
        assert torch.equal(model.engine.last["feat1d"], feat1d)
        assert torch.equal(fused, f2) and torch.equal(planes, p2) and torch.equal(centers, c2)
        # drift gate: the fixture's float figures against tests/golden/parity_baseline.json (common.drift_check)
        bad = drift_check("fixtures", case, report)
        assert not bad, "float parity drifted past the pinned baseline (re-pin with tools/update_parity_baseline.py " \
                        "in the same commit if intended): " + "; ".join(bad)
    finally:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(report) + "\n")
        print(report)


@pytest.mark.parametrize("case", ["panoptic_g_b2_thr", "campus_u_b2_all", "shelf_c_b1_thr"])
def test_proposal_layer_forward_standalone(case):
    """ProposalLayer.forward (human_detection_net.py:44-65, eval branch) called on its own, like the reference's module:
    fed with the reference's topk_index / topk_confs / match_bbox_preds it returns the reference's proposal_centers bit
    for bit (idx * scale + bias without fma, (conf > MIN_SCORE) - 1, pass-through columns); integer dtypes other than
    int64 are accepted; the training branch refuses."""
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case, device=DEV)
    model, _ = build(cfg, wseed, case)
    g = load_golden(case)
    want = g["proposal_centers_hdn"]                                             # HDN's output, before JLN rewrites [..., 4]
    X = cfg.CAPTURE_SPEC.VOXELS_PER_AXIS[0]
    flat = g["topk_flat"]
    idx = np.stack([flat // X, flat % X, np.argmax(g["hm1d"], axis=2)], axis=-1)  # get_index2D's divisor quirk (:13-33)
    layer = model.pose_net.proposal_layer
    for dt in (torch.int64, torch.int32):
        got = layer(torch.from_numpy(idx).to(DEV, dt), torch.from_numpy(want[..., 4]).to(DEV),
                    torch.from_numpy(np.ascontiguousarray(want[..., 5:7])).to(DEV), meta)
        assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)
    layer.train()
    with pytest.raises(NotImplementedError):
        layer(torch.from_numpy(idx).to(DEV), torch.from_numpy(want[..., 4]).to(DEV),
              torch.from_numpy(np.ascontiguousarray(want[..., 5:7])).to(DEV), {"roots_3d": 0, "num_person": 0})
    layer.eval()


@pytest.mark.parametrize("name", ["panoptic_b8", "shelf_b2", "panoptic128_b1", "campus_b2", "panoptic_b32"])
def test_float_parity_seed_sweep(name):
    """North-star float bar WITHOUT hand-picked seeds: 10 consecutive heatmap seeds per shape through the reference
    (tests/golden/make_seed_sweep.py), MIN_SCORE fixed at 0.4, Panoptic at the benchmark's own batch (B = 8), jln128
    and Campus included.  Rules R1 / R1p / R2 and their history in tests/golden/seed_sweep.py:
      R1  joints whose own reference fp32-vs-fp64 floor is <= 4e-4 mm: |build - ref32| <= 1e-3 mm - asserted with zero
          exceptions for panoptic_b8 and panoptic128_b1, reported for Shelf / Campus;
      R1p joints of proposals whose worst-joint floor is <= 4e-4 mm: <= 1e-3 mm, every shape;
      R1q joints with floor <= 4e-4 mm in proposals whose worst-joint floor is <= 1e-3 mm: <= 1e-3 mm - asserted for
          Panoptic, jln128 and Shelf (round 4), reported for Campus, whose asserted float bar is the conditioned fixture
          campus_c_b2_thr (tests/common.py FLOOR_RULE);
      R2  every joint: |build - ref32| <= 3 x max(its proposal's floor, 4e-4) - never noisier than the reference;
      proposal centres bit-equal; the overall fraction within 1e-3 mm goes to the parity report.
    Round 5: (i) panoptic_b32 - the largest batch bench.py quotes a rate for - with the Panoptic rules; (ii) Campus: the
    predicate of the conditioned Campus fixture (common.FLOOR_RULE: worst joint of a proposal within 2 x the proposal's own
    reference fp32-vs-fp64 floor) evaluated on EVERY compared proposal of all 10 seeds instead of one hand-picked seed:
    >= 95 % of the proposals must satisfy it and every one stays within 3 x (measured: 98 of 100, worst 2.77)."""
    import seed_sweep as SW
    s, rows = SW.replay(name, DEV, detail_path=os.path.join(os.path.dirname(REPORT), f"sweep_detail_{name}.npy"))
    rep = dict(s, case="sweep_" + name, rows=rows)
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(rep) + "\n")
    print(rep)
    assert s["seeds"] == SW.seeds_of(name) and s["joints"] > 300
    assert s["centres_exact"], "proposal centres / valid flags differ from the reference"
    if name in ("panoptic_b8", "panoptic128_b1", "panoptic_b32"):
        assert s["violations_where_floor_le_4e-4"] == 0, s                       # R1
    if name == "shelf_b2":
        # R1, literal, asserted for Shelf with ONE named exception instead of "reported" (VERDICT round 5, item 4): seed 6,
        # frame 0, slot 4, joint 4 - a joint with floor 3.9e-4 mm in a proposal whose own reference floor is 1.46e-3 mm.  It
        # read 1.008e-3 mm through the pixel-pair 7x7 form (rounds 3-5, and today under FVP_CONV_NO_K7 in the switch
        # matrix) and reads 9.86e-4 mm through k_conv7: the default build has no violation at all (pinned by the drift gate).
        det = np.load(os.path.join(os.path.dirname(REPORT), f"sweep_detail_{name}.npy"))
        viol = det[(det[:, 5] <= SW.FLOOR_OK) & (det[:, 4] > det[:, 7])]        # columns: seed, frame, slot, joint, err, floor, pfloor, bar
        named = {(6, 0, 4, 4)}
        got = {tuple(int(v) for v in r[:4]) for r in viol}
        assert got <= named, (got, viol[:, 4:7])
        for r in viol:
            assert r[6] >= 1.4e-3 and r[4] <= 1.05e-3, r                         # its proposal's own floor; barely over the bar
    assert s["violations_in_proposals_with_floor_le_4e-4"] == 0, s               # R1p
    if name != "campus_b2":
        assert s["violations_r1q"] == 0 and s["joints_r1q"] > 300, s             # R1q (Shelf: 910 joints in round 3)
    assert name == "campus_b2" or s["joints_of_proposals_with_floor_le_4e-4"] > 300
    assert s["worst_err_over_proposal_floor"] <= 3.0, s                          # R2
    # The factors are MEASURED, not chosen (round 6): tests/golden/reorder_distribution.json holds the same ratios for the
    # reference against ITSELF with another conv summation order (oneDNN on 8 threads vs ATen's native convs on one thread,
    # make_reorder_distribution.py).  The build may be as far from the reference as the reference is from itself, with a
    # margin of a quarter: worst proposal ratio <= 1.25 x the reference's own worst, and at least as many proposals within
    # 2 x their floor as the reference manages, less 2 %.
    mf = SW.measured_factors(name)
    if os.environ.get("FVP_TEST_DIAG_LIB") == "1":
        # switch-matrix rows run ALTERNATIVE kernels (other summation orders, e.g. the pixel-pair 7x7 form: Campus worst 2.77):
        # they keep the pre-round-6 constants - the measured bars describe the shipped kernel selection
        mf = {"joint_ratio_max": 3.0 / 1.25, "proposal_ratio_max": 3.0 / 1.25, "frac_within_2x": 0.97}
    assert s["worst_err_over_proposal_floor"] <= max(1.25 * mf["joint_ratio_max"], 1.0), (s, mf)   # R2 with the measured factor
    if name == "campus_b2":                                                      # FLOOR_RULE over the whole sweep
        assert s["proposals"] >= 90 and s["proposals_within_2x_own_floor"] >= (mf["frac_within_2x"] - 0.02) * s["proposals"], (s, mf)
        assert s["worst_proposal_err_over_own_floor"] <= 1.25 * mf["proposal_ratio_max"], (s, mf)
    bad = drift_check("sweeps", name, s)
    assert not bad, "float parity drifted past the pinned baseline (re-pin with tools/update_parity_baseline.py in the " \
                    "same commit if intended): " + "; ".join(bad)


def test_fused_projection_equals_materialised_full_size():
    """Panoptic jln64: fvp_project_individual_triplane == fvp_project_individual + fvp_triplane_max,
    bit for bit, and the drop-in ProjectLayer cubes have the reference's per-person checksums."""
    case = "panoptic_g_b1_all"
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case, device=DEV)
    g = load_golden(case)
    model, _ = build(cfg, wseed)
    heat, rt = heat.to(DEV), rt.to(DEV)
    with torch.no_grad():
        f1, p1, c1, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        planes_fused = model.engine.last_jln["planes"].clone()
        model.joint_net.fused_projection = False
        f2, p2, c2, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        assert torch.equal(model.engine.last_jln["planes"], planes_fused)
        assert torch.equal(f1, f2) and torch.equal(p1, p2) and torch.equal(c1, c2)
        pc = torch.from_numpy(g["proposal_centers_hdn"][0]).to(DEV)
        cubes, offset = model.joint_net.project_layer(heat, 0, meta, pc, cams, rt)
    np.testing.assert_allclose(cubes.double().sum(dim=(1, 2, 3, 4)).cpu().numpy(), g["jl0_cubes_sum"], rtol=1e-12)
    np.testing.assert_allclose((cubes.double() ** 2).sum(dim=(1, 2, 3, 4)).cpu().numpy(), g["jl0_cubes_sq"], rtol=1e-12)
    assert np.array_equal(offset.cpu().numpy(), g["jl0_offset"])


def test_batch_invariance_and_determinism_full_size():
    """Frames are independent units: a batch of 4 Panoptic frames equals the 4 single-frame runs
    bit for bit (the property the multi-GPU sharding relies on), and repeating a run is bit-stable."""
    cfg = S.make_cfg("panoptic", device=DEV, min_score=17.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).to(DEV)
    heat = S.heatmaps_blobs(cfg, cams, seq, 4, people=3, seed=21).to(DEV)
    model, _ = build(cfg)
    with torch.no_grad():
        fb, pb, cb, _, _ = model(meta={"seq": [seq] * 4}, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        fb2, _, _, _, _ = model(meta={"seq": [seq] * 4}, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        assert torch.equal(fb, fb2)
        assert 0 < int((cb[..., 3] >= 0).sum()) < 40, "fixture should mix valid and invalid proposals"
        for i in range(4):
            f1, p1, c1, _, _ = model(meta={"seq": [seq]}, input_heatmaps=heat[i:i + 1].contiguous(), cameras=cams,
                                     resize_transform=rt)
            assert torch.equal(f1[0], fb[i]) and torch.equal(c1[0], cb[i]) and torch.equal(p1[:, 0], pb[:, i])


def test_two_sequences_in_one_batch():
    """meta['seq'] may differ per frame: each frame uses its own camera set."""
    cfg = S.make_cfg("panoptic", device=DEV, min_score=-1.0)
    cams, seq = S.load_cameras("panoptic")
    import copy
    cams2 = copy.deepcopy(cams[seq])
    for c in cams2:
        c["T"] = [[c["T"][0][0] + 150.0], [c["T"][1][0] - 80.0], [c["T"][2][0]]]
    cameras = {seq: cams[seq], "shifted": cams2}
    rt = S.resize_transform(cfg).to(DEV)
    heat = S.heatmaps_blobs(cfg, cams, seq, 2, people=3, seed=5).to(DEV)
    model, _ = build(cfg)
    with torch.no_grad():
        fb, _, cb, _, _ = model(meta={"seq": [seq, "shifted"]}, input_heatmaps=heat, cameras=cameras, resize_transform=rt)
        f0, _, c0, _, _ = model(meta={"seq": [seq]}, input_heatmaps=heat[0:1].contiguous(), cameras=cameras, resize_transform=rt)
        f1, _, c1, _, _ = model(meta={"seq": ["shifted"]}, input_heatmaps=heat[1:2].contiguous(), cameras=cameras, resize_transform=rt)
    assert torch.equal(fb[0], f0[0]) and torch.equal(fb[1], f1[0])
    assert not torch.equal(cb[0], cb[1])


def test_empty_and_all_invalid():
    """No valid proposal at all (threshold above every confidence): zero joints, flags -1,
    HDN confidences kept; and an empty batch is accepted by every C-ABI call it reaches."""
    cfg = S.make_cfg("campus", device=DEV, min_score=1e9)
    cams, seq = S.load_cameras("campus")
    rt = S.resize_transform(cfg).to(DEV)
    heat = S.heatmaps_uniform(cfg, 2, seed=4).to(DEV)
    model, _ = build(cfg)
    with torch.no_grad():
        fused, planes, centers, _, _ = model(meta={"seq": [seq] * 2}, input_heatmaps=heat, cameras=cams, resize_transform=rt)
    assert torch.all(fused[..., :3] == 0) and torch.all(fused[..., 3] == -1) and torch.all(planes == 0)
    assert torch.equal(fused[..., 0, 4], centers[..., 4])


def test_nms_topk_exact_vs_oracle():
    from faster_voxelpose_amd.core.proposal import nms2D
    rng = np.random.default_rng(3)
    for (B, X, Y, N) in ((5, 80, 80, 10), (2, 128, 128, 10), (3, 16, 16, 7),
                         (2, 200, 200, 10), (2, 150, 130, 12)):     # > 16 384 cells: the LDS-resident form
        m = torch.from_numpy(rng.normal(size=(B, 1, X, Y)).astype(np.float32))
        m[0, 0, 0, 0] = m[0, 0, X - 1, Y - 1] = 9.0      # tie -> lowest index first
        m[1, 0, 3, 3] = m[1, 0, 3, 4] = 8.0              # plateau
        vals, idx, flat = nms2D(m.to(DEV), N)
        ov, oi, of = O.nms2d(m, N)
        assert torch.equal(flat.cpu(), of) and torch.equal(idx.cpu(), oi) and torch.equal(vals.cpu(), ov)


@pytest.mark.parametrize("cin,cmid,hw,planes", [(64, 128, (20, 20), 1), (128, 128, (20, 20), 8), (64, 64, (24, 20), 3),
                                                 (128, 64, (40, 40), 2)])
def test_split_k_direct_conv_on_small_maps(cin, cmid, hw, planes):
    """CenterNet's 3x3 layers on the 20x20 level (maps of 256 .. 576 pixels, >= 64 channels) run the split-K form of the
    direct kernel (four waves share a pixel block, fixed-order reduction); the 40x40 case runs masked Winograd tiles since
    round 6 (rows of >= 40 columns: netspec.WINO_MASKED_MIN_W).  Against a
    float64 torch evaluation, and bit-identical for a plane whatever the number of planes in the launch (the form is
    chosen from the layer shape, never from the batch)."""
    from faster_voxelpose_amd import _capi as capi
    lib = capi.load()
    spec, w, ref, o = split_k_stack(cin, cmid, hw, seed=cin)
    x = torch.from_numpy(np.random.default_rng(5).normal(size=(planes, cin) + hw).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    got = run_custom_conv_stack(lib, DEV, spec, w, x, st)[o].cpu()
    np.testing.assert_allclose(got.double().numpy(), ref(x).numpy(), rtol=2e-5, atol=2e-5)
    one = run_custom_conv_stack(lib, DEV, spec, w, x[planes - 1:], st)[o].cpu()
    assert torch.equal(one[0], got[planes - 1])


@pytest.mark.parametrize("cin,hw,planes", [(15, (64, 64), 240), (15, (80, 80), 8), (17, (64, 64), 31), (17, (80, 80), 3),
                                           (15, (10, 64), 3), (15, (128, 128), 5), (3, (4, 128), 1)])
def test_front_conv7_on_16x16x4_tiles(cin, hw, planes):
    """The 7x7 front conv of P2PNet / CenterNet at the BASELINE shapes (240 planes of 64x64, 8 of 80x80; 15 and 17 joints),
    heights that do not divide the 4-row tile, 128-wide maps.  Maps of 64 / 128 columns run k_conv7 (reduction ordered
    channel group / kernel row / kernel column on 16x16x4 tiles), 80 columns the pixel-pair form of k_conv_dma.  Against a
    float64 torch evaluation; a plane's bits do not depend on the number of planes in the launch nor on masked neighbours
    (the form is chosen from the layer shape only)."""
    from faster_voxelpose_amd import _capi as capi
    lib = capi.load()
    spec, w, ref, o = front7_stack(cin, hw, seed=cin)
    x = torch.from_numpy(np.random.default_rng(7).normal(size=(planes, cin) + hw).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    got = run_custom_conv_stack(lib, DEV, spec, w, x, st)[o].cpu()
    n = min(planes, 6)
    np.testing.assert_allclose(got[:n].double().numpy(), ref(x[:n]).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got[-1:].double().numpy(), ref(x[-1:]).numpy(), rtol=2e-5, atol=2e-5)
    one = run_custom_conv_stack(lib, DEV, spec, w, x[planes - 1:], st)[o].cpu()
    assert torch.equal(one[0], got[planes - 1])
    if planes > 2:
        valid = (torch.arange(planes) % 3 != 1).to(torch.uint8)
        masked = run_custom_conv_stack(lib, DEV, spec, w, x, st, plane_valid=valid)[o].cpu()
        assert torch.equal(masked[valid.bool()], got[valid.bool()])


@pytest.mark.parametrize("fused_head,head_cout", [(True, 15), (False, 15), (True, 17)])
def test_register_direct_conv_is_batch_independent(fused_head, head_cout):
    """1x1 convs / transposed convs (+ fused 1x1 head): 600 planes take k_conv_reg (>= 1024 tiles of 32 pixels on both map
    sizes), 3 planes take k_conv_dma - a plane's result must not depend on that (same MFMA chain: bit-identical), with a
    third of the planes masked out as well, and both match a float64 torch evaluation."""
    from faster_voxelpose_amd import _capi as capi
    lib = capi.load()
    spec, w, ref, outs = reg_stack(seed=4, fused_head=fused_head, head_cout=head_cout)
    x = torch.from_numpy(np.random.default_rng(6).normal(size=(600, 32, 16, 16)).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    big = run_custom_conv_stack(lib, DEV, spec, w, x, st)
    few = run_custom_conv_stack(lib, DEV, spec, w, x[:3], st)
    valid = (torch.arange(600) % 3 != 1).to(torch.uint8)
    masked = run_custom_conv_stack(lib, DEV, spec, w, x, st, plane_valid=valid)
    want = ref(x[:8])
    for name, o in outs.items():
        assert torch.equal(big[o][:3].cpu(), few[o].cpu()), name
        assert torch.equal(masked[o].cpu()[valid.bool()], big[o].cpu()[valid.bool()]), name
        np.testing.assert_allclose(big[o][:8].cpu().double().numpy(), want[name].numpy(), rtol=2e-5, atol=2e-5, err_msg=name)


def test_conv_stacks_vs_torch_fp32():
    """MFMA implicit-GEMM stacks vs the plain PyTorch fp32 restatement (oracle, CPU) at full
    Panoptic sizes, ragged plane counts included."""
    cfg = S.make_cfg("panoptic", device=DEV)
    model, sd = build(cfg)
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.random((7, 15, 64, 64), dtype=np.float32))
    want = O.p2p_net(sd, "joint_net.conv_net", x)
    got = model.joint_net.conv_net(x.to(DEV)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
    cubes = torch.from_numpy(rng.random((3, 15, 80, 80, 20), dtype=np.float32))
    hm_w, sz_w = O.center_net(sd, "pose_net.center_net", cubes)
    hm, sz = model.pose_net.center_net(cubes.to(DEV))
    np.testing.assert_allclose(hm.cpu().numpy(), hm_w.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sz.cpu().numpy(), sz_w.numpy(), rtol=2e-5, atol=2e-5)
    z = torch.from_numpy(rng.random((23, 15, 20), dtype=np.float32))
    np.testing.assert_allclose(model.pose_net.c2c_net(z.to(DEV)).cpu().numpy(),
                               O.c2c_net(sd, "pose_net.c2c_net", z).numpy(), rtol=2e-5, atol=2e-5)


def test_missing_sequence_and_wrong_device_fail_loudly():
    from faster_voxelpose_amd import _capi as capi
    cfg = S.make_cfg("campus", device=DEV)
    cams, seq = S.load_cameras("campus")
    rt = S.resize_transform(cfg).to(DEV)
    model, _ = build(cfg)
    heat = S.heatmaps_uniform(cfg, 1, seed=4)
    with pytest.raises(AssertionError):
        model(meta={"seq": ["nope"]}, input_heatmaps=heat.to(DEV), cameras=cams, resize_transform=rt)
    with pytest.raises(capi.FvpError):
        model(meta={"seq": [seq]}, input_heatmaps=heat, cameras=cams, resize_transform=rt)   # CPU tensor


def test_hipgraph_replay_equals_eager():
    """The captured hipGraph of the whole path reproduces the eager result bit for bit, also
    after the static input buffer is refilled with another batch."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("panoptic", device=DEV, min_score=17.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).to(DEV)
    h1 = S.heatmaps_blobs(cfg, cams, seq, 2, people=3, seed=31).to(DEV)
    h2 = S.heatmaps_blobs(cfg, cams, seq, 2, people=4, seed=32).to(DEV)
    model, _ = build(cfg)
    meta = {"seq": [seq] * 2}
    with torch.no_grad():
        e1 = [t.clone() for t in model(meta=meta, input_heatmaps=h1, cameras=cams, resize_transform=rt)[:3]]
        e2 = [t.clone() for t in model(meta=meta, input_heatmaps=h2, cameras=cams, resize_transform=rt)[:3]]
    g = FV.GraphedForward(model, meta, h1, cams, rt)
    o1 = [t.clone() for t in g(h1)[:3]]
    o2 = [t.clone() for t in g(h2)[:3]]
    for a, b in zip(e1 + e2, o1 + o2):
        assert torch.equal(a, b)


def test_fused_c2c_equals_generic_gpu():
    """One-kernel C2CNet (K split over four wave groups) vs the per-op interpreter at the Panoptic size (80
    columns): equal up to the rounding of the four-way partial sums; deterministic."""
    cfg = S.make_cfg("panoptic", device=DEV)
    model, sd = build(cfg)
    z = torch.from_numpy(np.random.default_rng(12).random((80, 15, 20), dtype=np.float32)).to(DEV)
    model.engine.fused_c2c = True
    fused = model.pose_net.c2c_net(z)
    model.engine.fused_c2c = False
    generic = model.pose_net.c2c_net(z)
    model.engine.fused_c2c = True
    np.testing.assert_allclose(fused.cpu().numpy(), generic.cpu().numpy(), rtol=3e-6, atol=3e-6)
    assert torch.equal(fused, model.pose_net.c2c_net(z))


@pytest.mark.gpu
def test_pipelined_forward_equals_plain_forward():
    """Three batches in flight on three streams (FV.PipelinedForward) give bit-identical results to
    the plain forward, batch by batch."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).to("cuda:0")
    model = FV.get(cfg).to("cuda:0")
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=11))
    heats = [S.heatmaps_blobs(cfg, cams, seq, 2, people=3, seed=40 + i).to("cuda:0") for i in range(5)]
    meta = {"seq": [seq] * 2}
    with torch.no_grad():
        want = []
        for h in heats:
            f, p, c, _, _ = model(meta=meta, input_heatmaps=h, cameras=cams, resize_transform=rt)
            want.append((f.clone(), p.clone(), c.clone()))
        torch.cuda.synchronize()
        pipe = FV.PipelinedForward(model, depth=3)
        got = [pipe.submit(meta=meta, input_heatmaps=h, cameras=cams, resize_transform=rt) for h in heats]
        pipe.synchronize()
    for (f, p, c), ((gf, gp, gc, _, _), ev) in zip(want, got):
        assert ev.query()
        assert torch.equal(f, gf) and torch.equal(p, gp) and torch.equal(c, gc)


@pytest.mark.gpu
def test_graphed_pipeline_equals_plain_forward():
    """FV.GraphedPipeline (one hipGraph per pipeline slot, static input buffer per slot): 10 batches through 3 slots equal
    the plain forward bit for bit; a slot's outputs stay valid until the slot comes round again."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).to("cuda:0")
    model = FV.get(cfg).to("cuda:0")
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=17))
    heats = [S.heatmaps_blobs(cfg, cams, seq, 2, people=3, seed=60 + i).to("cuda:0") for i in range(5)]
    meta = {"seq": [seq] * 2}
    with torch.no_grad():
        want = []
        for h in heats:
            f, p, c, _, _ = model(meta=meta, input_heatmaps=h, cameras=cams, resize_transform=rt)
            want.append((f.clone(), p.clone(), c.clone()))
        torch.cuda.synchronize()
        gp = FV.GraphedPipeline(model, 3, meta, heats[0], cams, rt)
        got = []
        for i in range(10):
            (f, p, c, _, _), ev = gp.submit(heats[i % 5])
            ev.synchronize()                             # (the slot's static outputs: copy before the slot is reused)
            got.append((f.clone(), p.clone(), c.clone()))
        gp.synchronize()
    for i, (f, p, c) in enumerate(got):
        w = want[i % 5]
        assert torch.equal(f, w[0]) and torch.equal(p, w[1]) and torch.equal(c, w[2]), i


@pytest.mark.gpu
def test_pipelined_single_frame_batches_are_bit_stable_under_concurrency():
    """B = 1 (half-size Winograd units, two workgroups per CU; direct kernels instead of the register-direct ones), four
    batches in flight, 40 batches: every result equals the serial forward bit for bit.  Kernels from other streams share
    the CUs here, which is where a timing-dependent hazard shows (round 5: a soft-argmax variant that was bit-exact alone
    differed in 1 of 4 batches under this load)."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).to("cuda:0")
    model = FV.get(cfg).to("cuda:0")
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=13))
    heats = [S.heatmaps_blobs(cfg, cams, seq, 1, people=4, seed=90 + i).to("cuda:0") for i in range(4)]
    meta = {"seq": [seq]}
    with torch.no_grad():
        want = []
        for h in heats:
            f, p, c, _, _ = model(meta=meta, input_heatmaps=h, cameras=cams, resize_transform=rt)
            want.append((f.clone(), p.clone(), c.clone()))
        torch.cuda.synchronize()
        pipe = FV.PipelinedForward(model, depth=4)
        got = [pipe.submit(meta=meta, input_heatmaps=heats[i % 4], cameras=cams, resize_transform=rt) for i in range(40)]
        pipe.synchronize()
    bad = [i for i, ((gf, gp, gc, _, _), _) in enumerate(got)
           if not (torch.equal(want[i % 4][0], gf) and torch.equal(want[i % 4][1], gp) and torch.equal(want[i % 4][2], gc))]
    assert not bad, f"batches that differ from the serial forward: {bad}"


@pytest.mark.gpu
def test_result_gather_over_rccl_is_host_issued_and_exact():
    """The N > 1 path on one GPU (RCCL process group of world size 1, `always=True` as bench.py's FVP_BENCH_FORCE_DIST): four
    batches in flight, every batch's fused poses all-gathered by the host-issued ResultGatherer (round 6: the collective is
    enqueued once the batch's event reports completion - no wait packet in any hardware queue).  The gathered rows equal the
    batch's own output bit for bit, in order; gathers are issued while later batches are still being submitted (the
    pipeline's back-pressure keeps the host at most four batches ahead); nothing is left pending after synchronize().
    Runs in its own process (tests/rccl_gather_check.py): RCCL prints a version banner through C stdio when the process
    exits, which would land behind pytest's summary line."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_gather_check.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL GATHER OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_pipelined_batches_each_introducing_a_new_sequence():
    """Consecutive in-flight batches on different streams each bring a NEW sequence: the second rebuild of the shared
    camera table / coordinate cache reads what the first is still writing on another stream (ADVICE round 4:
    SharedGeometry kept only the LAST rebuild's event).  Every batch must equal the plain forward of a model that has
    seen all sequences up front, and the superseded tables must be released once every stream has passed them."""
    import copy
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).to("cuda:0")
    cameras = {seq: cams[seq]}
    names = [seq]
    for i in range(1, 6):
        c2 = copy.deepcopy(cams[seq])
        for c in c2:
            c["T"] = [[c["T"][0][0] + 60.0 * i], [c["T"][1][0] - 35.0 * i], [c["T"][2][0]]]
        cameras[f"shift{i}"] = c2
        names.append(f"shift{i}")
    heats = [S.heatmaps_blobs(cfg, cams, seq, 2, people=3, seed=70 + i).to("cuda:0") for i in range(len(names))]
    ref = FV.get(cfg).to("cuda:0")
    sd = S.fill_state_dict(ref.state_dict(), seed=11)
    ref.load_state_dict(sd)
    model = FV.get(cfg).to("cuda:0")
    model.load_state_dict(sd)
    with torch.no_grad():
        want = []
        for n, h in zip(names, heats):
            f, p, c, _, _ = ref(meta={"seq": [n, names[0]]}, input_heatmaps=h, cameras=cameras, resize_transform=rt)
            want.append((f.clone(), p.clone(), c.clone()))
        torch.cuda.synchronize()
        pipe = FV.PipelinedForward(model, depth=4)
        got = [pipe.submit(meta={"seq": [n, names[0]]}, input_heatmaps=h, cameras=cameras, resize_transform=rt)
               for n, h in zip(names, heats)]
        pipe.synchronize()
        torch.cuda.synchronize()
    for (f, p, c), ((gf, gp, gc, _, _), ev) in zip(want, got):
        assert ev.query()
        assert torch.equal(f, gf) and torch.equal(p, gp) and torch.equal(c, gc)
    geo = model.engine.geo
    assert geo.cams.shape[0] == len(names) and geo.fine_grid.shape[0] == len(names)
    geo.retire(None)                                 # everything has drained: nothing superseded stays alive
    assert geo.retired == [] and len(geo.users) >= 2


@pytest.mark.gpu
def test_standalone_softargmax_and_weightnet_vs_oracle():
    """SoftArgmaxLayer.forward / WeightNet.forward as standalone launches (reference layout
    [3,P,J,C,C]) against the oracle's restatement of joint_localization_net.py:20-34 and
    weight_net.py:69-80."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
    model = FV.get(cfg).to("cuda:0")
    sd = S.fill_state_dict(model.state_dict(), seed=5)
    model.load_state_dict(sd)
    P, J, Cn = 4, cfg.DATASET.NUM_JOINTS, cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS[0]
    rng = np.random.default_rng(3)
    jn = model.joint_net
    grid = model.engine.center_grid
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    # (a) single-mode maps (what a trained P2PNet emits): one bump of height 0.3 and sigma 2 cells per
    #     map on a 0.02 noise background -> the oracle's fp32 result is well defined: 1e-3 mm bar
    yy, xx = np.mgrid[0:Cn, 0:Cn]
    cx, cy = rng.uniform(8, Cn - 8, (2, 3, P, J, 1, 1))
    bump = 0.3 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 2.0 ** 2)) + 0.02 * rng.random((3, P, J, Cn, Cn))
    # (b) uniform noise in [0, 0.2): almost flat softmax, the fp32 expectation of the oracle itself
    #     carries ~1e-2 mm of rounding noise -> compare with the float64 expectation instead
    flat = rng.random((3, P, J, Cn, Cn)) * 0.2
    for x, acc in ((torch.from_numpy(bump.astype(np.float32)), torch.float32),
                   (torch.from_numpy(flat.astype(np.float32)), torch.float64)):
        with torch.no_grad():
            pose, confs = jn.soft_argmax_layer(x.cuda(), grid)
            w = jn.weight_net(x.cuda())
        want_pose, want_conf = O.soft_argmax(x, grid.cpu(), float(cfg.NETWORK.BETA), accumulate=acc)
        want_w = O.weight_net(sd_cpu, "joint_net.weight_net", x, Cn)
        assert pose.shape == (3, P, J, 2) and confs.shape == (P,) and w.shape == (3 * P, J, 1)
        np.testing.assert_allclose(pose.cpu().numpy(), want_pose.numpy(), rtol=0, atol=1e-3)     # mm
        np.testing.assert_allclose(confs.cpu().numpy(), want_conf.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(w.cpu().numpy(), want_w.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["hm_shelf_p4", "hm_campus_p3", "hm_panoptic_p6"])
def test_rasteriser_vs_reference_golden(case):
    """GPU heatmap rasteriser (fvp_rasterise_heatmaps) vs the reference's generate_input_heatmap:
    windows are placed with exact float64 arithmetic, values are exp(double) rounded once to fp32
    (bit-equal up to the device libm's last float64 bit: <= 1 ulp of float32 allowed)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from heatmap_cases import make_pred2d
    from faster_voxelpose_amd.dataset import generate_input_heatmaps
    cfg, all_preds, rt, sigma = make_pred2d(case)
    cfg.DEVICE = "cuda:0"
    g = load_golden(case)["heatmaps"]
    hm, cl = generate_input_heatmaps(all_preds, rt, cfg, sigma=sigma, channels_last=True)
    got = hm.cpu().numpy()
    assert np.array_equal(got > 0, g > 0), "window placement differs"
    np.testing.assert_allclose(got, g, rtol=1.2e-7, atol=1e-45)
    J = cfg.DATASET.NUM_JOINTS
    V, _, H, W = g.shape
    assert np.array_equal(cl.cpu().numpy()[..., :J], got.reshape(V, J, H * W).transpose(0, 2, 1))
    # batched call = per-frame calls
    hm2 = generate_input_heatmaps([all_preds, all_preds], rt, cfg, sigma=sigma)
    assert torch.equal(hm2[0], hm) and torch.equal(hm2[1], hm)


@pytest.mark.gpu
def test_precomputed_heatmap_path_end_to_end_shelf():
    """BASELINE configs[2] shape (Shelf, 'pred' heatmap source): 2-D detections -> GPU rasteriser ->
    hot path -> PCP evaluator, through core.function.validate_batches; compared with the CPU oracle fed with
    the oracle's own rasterised heatmaps."""
    import functools
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from heatmap_cases import make_pred2d
    from faster_voxelpose_amd.core import function as FN, metrics as M
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg, _, rt, _ = make_pred2d("hm_shelf_p4")
    sigma = 2.0
    cfg.DEVICE = "cuda:0"
    cfg.CAPTURE_SPEC.MIN_SCORE = -1.0
    cfg.NETWORK.SIGMA = sigma
    cams, seq = S.load_cameras("shelf")
    # detections = projections of 4 consistent 3-D skeletons (every view sees every person)
    all_preds = S.pred2d_people(cfg, cams, seq, 4, seed=4, region=1400.0, spacing=1400.0, joint_std=(120.0, 120.0, 250.0))
    model = FV.get(cfg).to("cuda:0")
    sd = S.fill_state_dict_conditioned(model.state_dict(), seed=11)
    model.load_state_dict(sd)
    batches = [dict(meta={"seq": [seq, seq]}, pred_pose2d=[all_preds, all_preds]),
               dict(meta={"seq": [seq]}, pred_pose2d=[all_preds])]
    actors = [[np.random.default_rng(a).normal(0, 500, (14, 3)) for _ in range(3)] for a in range(4)]
    metric, fused, info = FN.validate_batches(cfg, model, batches, cams, rt, depth=2,
                                      evaluate=functools.partial(M.evaluate_pcp, actors_mm=actors))
    assert fused.shape == (3, cfg.CAPTURE_SPEC.MAX_PEOPLE, cfg.DATASET.NUM_JOINTS, 5) and info["frames"] == 3
    assert 0.0 <= metric <= 1.0 and set(info["evaluation"]) >= {"actor_pcp", "recall"}
    assert torch.equal(fused[0], fused[1]) and torch.equal(fused[0], fused[2])       # same frame three times
    # oracle: its own rasteriser + its own pipeline, in fp32 and with the joint net in float64
    cfg_cpu = S.make_cfg("shelf", device="cpu", min_score=-1.0)
    hm = O.input_heatmaps_from_pred2d(all_preds, rt, cfg_cpu.DATASET.IMAGE_SIZE, cfg_cpu.DATASET.HEATMAP_SIZE, sigma)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    rt32 = torch.as_tensor(rt, dtype=torch.float32)
    of, _, oc = O.Oracle(cfg_cpu, sd_cpu).forward(hm[None], {"seq": [seq]}, cams, rt32)
    of64, _, _ = O.Oracle(cfg_cpu, sd_cpu).forward(hm[None], {"seq": [seq]}, cams, rt32, net_dtype=torch.float64)
    assert torch.equal(fused[0].cpu()[..., 3], of[0][..., 3])
    err = (fused[0].cpu()[..., :3] - of[0][..., :3]).norm(dim=-1).max(dim=1)[0]           # per proposal
    floor = (of[0][..., :3] - of64[0][..., :3].float()).norm(dim=-1).max(dim=1)[0]
    # proposals whose joint maps are well conditioned by the oracle's own measure (fp32 vs fp64 within
    # 4e-4 mm: the real people): the 1e-3 mm bar; ghost proposals (all valid with MIN_SCORE = -1): 3x floor
    good = floor <= 4e-4
    assert int(good.sum()) >= 3, floor
    assert float(err[good].max()) <= 1e-3, (err, floor)
    assert bool((err[~good] <= 3 * floor[~good] + 1e-3).all()), (err, floor)


@pytest.mark.gpu
def test_backbone_bf16_vs_oracle_and_reference_golden():
    """Pose-ResNet-50 on the GPU (fvp_bb_run, bf16 MFMA) on the golden's [2,3,96,128] batch: as close
    to the reference's fp32 heatmaps as a bf16 evaluation gets (the oracle's bf16-emulating mode is
    the yardstick), identical NCHW / channels-last outputs, and drop-in use as `backbone=` of the
    voxel model."""
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import sys
    sys.path.insert(0, sys_path)
    from make_golden_backbone import WSEED, inputs
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    g = load_golden("backbone_r50")
    cfg = CFG.default_config()
    m = RN.get(cfg).to("cuda:0")
    sd = S.fill_backbone_state_dict(m.state_dict(), seed=WSEED)
    m.load_state_dict(sd)
    x = inputs()
    with torch.no_grad():
        y = m(x.cuda())
        cl = m.forward_channels_last(x.cuda())
    ref = torch.from_numpy(g["heatmaps"])
    o16 = O.pose_resnet(sd, x, bf16=True)
    e_prod = float((y.cpu() - ref).norm() / ref.norm())
    e_orc = float((o16 - ref).norm() / ref.norm())
    assert e_prod < 1.5 * e_orc + 1e-3, (e_prod, e_orc)
    J = cfg.DATASET.NUM_JOINTS
    assert torch.equal(cl[..., :J], y.reshape(2, J, -1).permute(0, 2, 1)) and not cl[..., J:].any()
    assert torch.equal(y, m(x.cuda()))                       # deterministic


@pytest.mark.gpu
def test_end_to_end_images_to_joints_config5_shape():
    """BASELINE configs[4] shape in miniature: images -> bf16 backbone -> voxel pipeline in one forward
    (`views=` + `backbone=`, like run/validate.py with TEST_HEATMAP_SRC = 'image').  The heatmaps the model
    returns equal a standalone backbone call; the joints equal a forward on those heatmaps (the adopted
    channels-last copy and the restaged one are the same data)."""
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import faster_voxelpose as FV, resnet as RN
    cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
    cams, seq = S.load_cameras("panoptic")
    rt = S.resize_transform(cfg).cuda()
    model = FV.get(cfg).to("cuda:0")
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
    bcfg = CFG.default_config()
    bb = RN.get(bcfg).to("cuda:0")
    bb.load_state_dict(S.fill_backbone_state_dict(bb.state_dict(), seed=3))
    W, H = cfg.DATASET.IMAGE_SIZE
    views = torch.rand(1, 5, 3, H, W, device="cuda")
    meta = {"seq": [seq]}
    with torch.no_grad():
        fused, planes, centers, heat, _ = model(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt)
        heat2 = bb(views[0])
        assert heat.shape == (1, 5, cfg.DATASET.NUM_JOINTS, H // 4, W // 4) and torch.equal(heat[0], heat2)
        f2, p2, c2, _, _ = model(meta=meta, input_heatmaps=heat.clone(), cameras=cams, resize_transform=rt)
    assert torch.equal(fused, f2) and torch.equal(centers, c2)
    assert torch.isfinite(fused).all()
    # the same through three batches in flight sharing the one backbone instance
    pipe = FV.PipelinedForward(model, depth=3)
    outs = [pipe.submit(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt) for _ in range(4)]
    pipe.synchronize()
    for (pf, _, pc, ph, _), _ in outs:
        assert torch.equal(pf, fused) and torch.equal(pc, centers) and torch.equal(ph, heat)


@pytest.mark.gpu
def test_backbone_full_image_size_vs_oracle():
    """Two 512 x 960 images (the Panoptic network-image size: pixel indices in the millions) against the
    CPU oracle: same closeness to fp32 as the oracle's bf16-emulating evaluation."""
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    cfg = CFG.default_config()
    m = RN.get(cfg).to("cuda:0")
    sd = S.fill_backbone_state_dict(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    x = torch.from_numpy(np.random.default_rng(9).random((2, 3, 512, 960), dtype=np.float32))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
        o32 = O.pose_resnet(sd, x)
        o16 = O.pose_resnet(sd, x, bf16=True)
    e_prod = float((y - o32).norm() / o32.norm())
    e_orc = float((o16 - o32).norm() / o32.norm())
    assert e_prod < 1.5 * e_orc + 1e-3, (e_prod, e_orc)
    # per-image agreement too (an indexing slip would hit the second image)
    for n in range(2):
        assert float((y[n] - o32[n]).norm() / o32[n].norm()) < 1.5 * e_orc + 2e-3


@pytest.mark.gpu
def test_backbone_tile_shapes_give_identical_heatmaps(monkeypatch, diag_lib):
    """The LDS-DMA conv kernel with 256-cout tiles where they fill the chip (default), with 128-cout tiles everywhere
    and at a batch that leaves partial pixel tiles / partial XCD groups: the tile shape changes neither the k order
    nor the MFMA shape, so the heatmaps are bit-identical.  The register-staged 128 x 128 kernel walks k tap-major
    instead: equal up to bf16 rounding flips."""
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    cfg = CFG.default_config()
    # the switches below are honoured by the diagnostics build only (tests/diag; the shipped library reads no environment)
    m = RN.PoseResNet(cfg, _lib=diag_lib).to("cuda:0")
    m.load_state_dict(S.fill_backbone_state_dict(m.state_dict(), seed=5))
    x = torch.from_numpy(np.random.default_rng(2).random((3, 3, 160, 224), dtype=np.float32)).cuda()
    with torch.no_grad():
        y0 = m(x).clone()
        mp = RN.get(cfg).to("cuda:0")                      # the product library: same bits as the diagnostics build's default
        mp.load_state_dict(m.state_dict())
        assert torch.equal(mp(x), y0)
        monkeypatch.setenv("FVP_BB_DMA_BN", "128")
        y1 = m(x).clone()
        monkeypatch.delenv("FVP_BB_DMA_BN")
        monkeypatch.setenv("FVP_BB_NO_FUSE_FINAL", "1")      # heatmap layer as its own kernel
        y3 = m(x).clone()
        cl3 = m.forward_channels_last(x).clone()
        monkeypatch.delenv("FVP_BB_NO_FUSE_FINAL")
        cl0 = m.forward_channels_last(x).clone()
        monkeypatch.setenv("FVP_BB_NO_BIG", "1")
        y2 = m(x).clone()
    assert torch.equal(y0, y1)
    # fused heatmap layer: the same bf16 products, summed per cout half and then across -> fp32 rounding only
    assert torch.allclose(y0, y3, rtol=1e-4, atol=2e-6) and torch.allclose(cl0, cl3, rtol=1e-4, atol=2e-6)
    assert float((y2 - y0).norm() / y0.norm()) < 2e-2 and not torch.equal(y2, torch.zeros_like(y2))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 160, 224), (2, 512, 960), (1, 96, 64)])
def test_backbone_fused_kernels_equal_the_separate_launches(monkeypatch, diag_lib, shape):
    """Round 6: the stem conv + bn + ReLU + max-pool kernel (k_bb_stem_pool) and the fused bottleneck kernels against the
    layer-by-layer launches they replace (FVP_BB_NO_FUSE_STEM / FVP_BB_NO_FUSE_BLOCK, diagnostics build).  Each fused
    kernel runs the MFMA chain of the layers it replaces in the same k order and rounds to bf16 at the same points, so
    the heatmaps are bit-equal - at the Panoptic image size, at a size with ragged tiles and on a small image."""
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    cfg = CFG.default_config()
    m = RN.PoseResNet(cfg, _lib=diag_lib).to("cuda:0")
    m.load_state_dict(S.fill_backbone_state_dict(m.state_dict(), seed=5))
    m.autotune = False
    n, h, w = shape
    x = torch.from_numpy(np.random.default_rng(4).random((n, 3, h, w), dtype=np.float32)).cuda()
    with torch.no_grad():
        fused = m(x).clone()
        mp = RN.get(cfg).to("cuda:0")                      # the product library runs the fused kernels too: same bits
        mp.load_state_dict(m.state_dict())
        mp.autotune = False
        assert torch.equal(mp(x), fused)
        monkeypatch.setenv("FVP_BB_NO_FUSE_STEM", "1")
        no_stem = m(x).clone()                             # stem + pooling as two launches, bottlenecks fused
        monkeypatch.setenv("FVP_BB_NO_FUSE_BLOCK", "1")
        plain = m(x).clone()                               # every layer its own launch
    assert float(fused.abs().max()) > 0
    assert torch.equal(fused, no_stem), ("stem", float((fused - no_stem).abs().max()))
    assert torch.equal(fused, plain), ("bottleneck", float((fused - plain).abs().max()))


@pytest.mark.gpu
def test_backbone_tile_configurations_are_bit_identical_per_op():
    """fvp_bb_tune picks one of three tile configurations of the LDS-DMA conv kernel per op by timing them.  That is
    only legitimate if the choice cannot change a result: every eligible op of the Pose-ResNet-50 plan, run alone
    on the same random bf16 inputs with each configuration, gives identical bits."""
    import ctypes as C
    from faster_voxelpose_amd import _capi as capi
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    m = RN.get(CFG.default_config()).to("cuda:0")
    m.load_state_dict(S.fill_backbone_state_dict(m.state_dict(), seed=5))
    m.autotune = False
    N, H, W = 3, 160, 224
    with torch.no_grad():
        m(torch.rand(N, 3, H, W, device="cuda"))                       # packs the weights
    plan = m._plan(H, W)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    bufs = []
    for name in plan["names"]:
        c, h, w = plan["shapes"][name]
        bufs.append((torch.rand((N, h, w, c), device="cuda", generator=g) - 0.5).bfloat16())
    arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    checked = 0
    for op in plan["ops"]:
        if op.kind == capi.BB_MAXPOOL or op.dst < 0 or op.cinp % 64 or op.coutp % 128:
            continue
        outs = []
        for cfgv in ((1, 2, 3) if op.coutp % 256 == 0 else (2, 3)):
            one = (capi.FvpBbOp * 1)(op)
            one[0].flags = (op.flags & ~(3 << 8)) | (cfgv << 8)
            bufs[op.dst].zero_()
            capi.check(m.lib, m.lib.fvp_bb_run(one, 1, C.c_void_p(m._wblob.data_ptr()), C.c_void_p(m._eblob.data_ptr()), arr,
                                               len(bufs), N, None, 0, None, s), "fvp_bb_run")
            torch.cuda.synchronize()
            outs.append(bufs[op.dst].clone())
        assert outs[0].float().abs().max() > 0
        assert all(torch.equal(outs[0], o) for o in outs[1:]), (op.cin, op.cout, op.kh, op.h, op.w)
        checked += 1
    assert checked >= 40
    # and the tuner itself leaves the heatmaps untouched
    x = torch.rand(N, 3, H, W, device="cuda")
    with torch.no_grad():
        y0 = m(x).clone()
        m.autotune = True
        y1 = m(x).clone()
    assert N in plan["tuned"] and torch.equal(y0, y1)


# ---- edge cases shared with the emulator suite (tests/edge_cases.py) ---------------------------------
@pytest.mark.gpu
def test_backbone_single_layers_vs_bf16_oracle():
    """Per-layer parity of the bf16 backbone kernels on the GPU (layers were only compared on
    the emulator): the first conv of every stage, every stride-2 3x3, every downsample, a residual expansion per stage
    and all three transposed convs of the Pose-ResNet-50 plan, each run ALONE through fvp_bb_run on random bf16
    activations and compared with the oracle's numerics for that layer (bf16 inputs / weights, fp32 accumulation, eval
    BatchNorm as scale / shift, residual, ReLU, one rounding to bf16).  The two differ only in the fp32 summation
    order in front of the bf16 rounding: every element within one bf16 ulp (2e-6 absolute where a layer's terms cancel),
    all but a few per cent bit-equal."""
    import ctypes as C
    import torch.nn.functional as F
    from faster_voxelpose_amd import _capi as capi
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    m = RN.get(CFG.default_config()).to("cuda:0")
    sd = S.fill_backbone_state_dict(m.state_dict(), seed=11)
    m.load_state_dict(sd)
    m.autotune = False
    N, H, W = 2, 128, 160
    with torch.no_grad():
        m(torch.rand(N, 3, H, W, device="cuda"))                       # packs the weights
    plan = m._plan(H, W)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    bufs = []
    for name in plan["names"]:
        c, h, w = plan["shapes"][name]
        bufs.append((torch.rand((N, h, w, c), device="cuda", generator=g) - 0.5).bfloat16())
    arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    want = set()
    for li in (1, 2, 3, 4):
        want |= {f"layer{li}.0.conv1", f"layer{li}.0.conv2", f"layer{li}.0.downsample.0", f"layer{li}.0.conv3", f"layer{li}.1.conv3"}
    want |= {"deconv_layers.0", "deconv_layers.3", "deconv_layers.6"}
    checked, worst_frac = 0, 0.0
    for i, o in enumerate(m._convs):
        if o.get("key") not in want:
            continue
        op = plan["ops"][i]
        one = (capi.FvpBbOp * 1)(op)
        bufs[op.dst].zero_()
        capi.check(m.lib, m.lib.fvp_bb_run(one, 1, C.c_void_p(m._wblob.data_ptr()), C.c_void_p(m._eblob.data_ptr()), arr,
                                           len(bufs), N, None, 0, None, st), "fvp_bb_run")
        torch.cuda.synchronize()
        got = bufs[op.dst].float().cpu()                                           # NHWC
        x = bufs[op.src].float().cpu().permute(0, 3, 1, 2)
        wq = sd[o["key"] + ".weight"].bfloat16().float()
        if op.kind == capi.BB_DECONV:
            y = F.conv_transpose2d(x, wq, None, stride=2, padding=1)
        else:
            y = F.conv2d(x, wq, None, stride=o["stride"], padding=o["pad"])
        sc = sd[o["bn"] + ".weight"] / torch.sqrt(sd[o["bn"] + ".running_var"] + 1e-5)
        sh = sd[o["bn"] + ".bias"] - sd[o["bn"] + ".running_mean"] * sc
        if o.get("bias"):
            sh = sh + sd[o["key"] + ".bias"] * sc
        y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        if op.res >= 0:
            y = y + bufs[op.res].float().cpu().permute(0, 3, 1, 2)
        if o.get("relu"):
            y = F.relu(y)
        ref = y.bfloat16().float().permute(0, 2, 3, 1)
        assert ref.shape == got.shape, (o["key"], ref.shape, got.shape)
        diff = (got - ref).abs()
        # one bf16 ulp (8 significand bits); where the layer's terms cancel to ~0 the two fp32 summation orders differ by
        # ~2^-24 of the summed magnitudes instead (absolute floor 2e-6)
        ulp = torch.maximum(ref.abs(), got.abs()) * 2.0 ** -7 + 2e-6
        if not bool((diff <= ulp).all()):
            bad = torch.nonzero(diff > ulp)
            print(o["key"], "bad elements", len(bad), "of", diff.numel(), "first:", [(tuple(int(v) for v in b), float(got[tuple(b)]), float(ref[tuple(b)])) for b in bad[:12]])
            print("   bad channels", sorted(set(int(b[3]) for b in bad))[:40], "bad rows", sorted(set(int(b[1]) for b in bad))[:20])
        assert bool((diff <= ulp).all()), (o["key"], float(diff.max()), float((diff / ulp).max()))
        frac = float((diff > 0).float().mean())
        worst_frac = max(worst_frac, frac)
        assert frac < 0.05, (o["key"], frac)
        assert float(got.abs().max()) > 0
        checked += 1
    print(f"backbone single layers: {checked} ops, worst mismatching fraction {worst_frac:.4f} (each within one bf16 ulp)")
    assert checked == len(want)


@pytest.mark.gpu
def test_zero_batch_through_every_export():
    import edge_cases as E
    E.zero_batch_through_every_export(None, DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["tiny", "panoptic"])
def test_negative_bbox_gives_an_empty_window(shape):
    import edge_cases as E
    E.negative_bbox_gives_an_empty_window(None, DEV, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["panoptic", "shelf", "campus"])
def test_sampling_grids_equal_reference(shape):
    """a-2 (project_grid / project_point / the 2x3 affine) on the GPU: fvp_sample_grid against the digests
    of the reference's cached whole-space and fine grids, bit for bit."""
    import edge_cases as E
    E.sampling_grids_equal_reference(None, DEV, shape)


@pytest.mark.gpu
def test_checkpoint_file_drives_the_gpu_model(tmp_path):
    """A model_best.pth.tar-style file (bare state_dict, utils.py:92-98) and a train.py-style full
    checkpoint (DataParallel prefix, backbone entries, np.float64 precision) loaded through
    utils/checkpoint.py give the same joints as load_state_dict of the tensors themselves."""
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    from faster_voxelpose_amd.utils import checkpoint as CK
    case = "panoptic_c_b2_thr"
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case, device=DEV)
    model, sd = build(cfg, wseed, case)
    with torch.no_grad():
        want = model(meta=meta, input_heatmaps=heat.to(DEV), cameras=cams, resize_transform=rt.to(DEV))[0].clone()
    best = tmp_path / "model_best.pth.tar"
    torch.save({k: v.cpu() for k, v in sd.items()}, best)
    full = tmp_path / "checkpoint.pth.tar"
    torch.save({"epoch": 3, "precision": np.float64(0.9), "state_dict": {
        **{"module." + k: v.cpu() for k, v in sd.items()}, "module.backbone.conv1.weight": torch.zeros(4)}}, full)
    for path in (best, full):
        m = FV.get(cfg).to(DEV)
        rep = CK.load_model_file(m, str(path))
        assert rep == dict(missing=[], unexpected=[], shape_mismatch=[])
        with torch.no_grad():
            got = m(meta=meta, input_heatmaps=heat.to(DEV), cameras=cams, resize_transform=rt.to(DEV))[0]
        assert torch.equal(got, want)
