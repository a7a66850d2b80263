"""Shared comparison logic: product outputs (HIP on the GPU, or the emulated kernels on the
CPU) against the golden vectors captured from the reference."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star bar: 3-D joints within 1e-3 mm of the reference.
#  * conditioned fixtures (flavour "c" in cases.py: trained-like single-mode joint maps; the
#    reference's own fp32-vs-fp64 floor is <= ~4e-4 mm): |build - reference| <= 1e-3 mm, no escape.
#  * stress fixtures (random weights -> multi-modal joint maps with near-ties, uniform-noise
#    heatmaps): the reference's fp32 result itself is only reproducible to its rounding-noise floor
#    (fp32 vs the same nets in fp64, stored per fixture as margins[5]: 3e-3 .. 9e-2 mm), and any other
#    summation order inside the convs lands anywhere within that floor.  There: pass at <= 1e-3 mm, or
#    when |build - ref32| <= FLOOR_FACTOR x floor AND the distance to the float64 evaluation
#    |build - ref64| is <= 1e-3 mm -- except for the cases listed in FP64_EXCEPTIONS.
BAR_MM = 1e-3
FLOOR_FACTOR = 3.0
# |build - ref64| bound (mm) where it is not <= 1e-3: the three uniform-noise heatmap fixtures.  Their
# joint maps are almost flat (softmax mass spread over the whole 2 m cube, mean absolute deviation
# ~500 mm), so the build's own fp32 conv rounding (|d feat| ~ 1e-7, times beta = 100, times 500 mm)
# moves the expectation by a few 1e-3 mm, exactly like the reference's (floors 5e-3 .. 5e-2 mm there).
FP64_EXCEPTIONS = {"tiny_u_b3_thr": 3e-3, "campus_u_b2_all": 4e-3, "panoptic_u_b1_all": 6e-3}


# Conditioned fixtures on which the REFERENCE's fp32 output is itself further than 1e-3 mm from its own float64 evaluation,
# so "within 1e-3 mm of the reference's fp32 result" is below what the reference reproduces.  Campus: 3 views, world
# coordinates of 4 .. 7.5 m (one fp32 ulp = 4.9e-4 mm), blurry 360 x 288 images -> every person of every seed has a
# reference fp32-vs-fp64 distance of 0.8e-3 .. 2.0e-3 mm (tests/golden/find_conditioned.py, cases.py).  The bar asserted
# there, per valid proposal p with pfloor_p = max_j |ref32 - ref64| (the reference's own error on that person):
#     max_j |build - ref64| <= K64 * pfloor_p     the build is at most 1.5x as far from the exact answer as the reference
#     max_j |build - ref32| <= K32 * pfloor_p     and within twice the reference's own error of the reference's fp32 result
# (tighter than the triangle inequality K64 + 1 = 2.5, and than the seed sweep's R2 factor 3).  Fixed before the first GPU
# run of this fixture; the CPU emulation of the kernels gave worst ratios 1.26 / 1.71 (9 proposals, pfloor 0.96e-3 .. 1.97e-3,
# |build - ref32| max 2.0e-3 mm: the literal 1e-3 mm bar is NOT met on Campus, by the reference's fp32 path either).
# Round 6: K32 is no longer only a chosen constant - the reference run against ITSELF with another conv summation order
# (tests/golden/reorder_distribution.json, make_reorder_distribution.py) lands up to 1.62 x a Campus proposal's own floor
# from its fp32 result (p95 1.38, 100 of 100 proposals within 2 x): K32 = 2.0 = 1.25 x that maximum, rounded.
FLOOR_RULE = {"campus_c_b2_thr": (1.5, 2.0)}
FLOOR_RULE_ABS_MM = 2.5e-3


def floor_rule_check(case, xyz, g, report=None):
    """Assert FLOOR_RULE[case] for joints ``xyz`` [B,N,J,3] (numpy) against the golden ``g``; returns the worst ratios."""
    k64, k32 = FLOOR_RULE[case]
    v = g["valid"]
    f3 = xyz.astype(np.float64)
    d32 = np.linalg.norm(f3 - g["fused_poses"][..., :3], axis=-1)            # [B,N,J]
    d64 = np.linalg.norm(f3 - g["floor_fused"], axis=-1)
    pfl = np.linalg.norm(g["fused_poses"][..., :3].astype(np.float64) - g["floor_fused"], axis=-1).max(axis=-1)   # [B,N]
    r64 = d64.max(axis=-1)[v] / pfl[v]
    r32 = d32.max(axis=-1)[v] / pfl[v]
    if report is not None:
        report.update(worst_ratio_vs_fp64=float(r64.max()), worst_ratio_vs_ref32=float(r32.max()),
                      proposal_floor_min_mm=float(pfl[v].min()), proposal_floor_max_mm=float(pfl[v].max()))
    assert pfl[v].min() > 5e-4, "fixture no longer needs the floor rule"
    # an absolute ceiling beside the ratio bars: a regression that scales with the proposals' own floors
    # cannot hide behind them (measured on the MI355X: 1.97e-3 mm)
    assert d32.max(axis=-1)[v].max() <= FLOOR_RULE_ABS_MM, f"{case}: |build-ref32| max {d32.max(axis=-1)[v].max():.2e} mm"
    assert r64.max() <= k64 and r32.max() <= k32, \
        f"{case}: |build-ref64| / pfloor max {r64.max():.2f} (bar {k64}), |build-ref32| / pfloor max {r32.max():.2f} (bar {k32})"
    return float(r64.max()), float(r32.max())


# ---- drift gate ---------------------------------------------------------------------------
# tests/golden/parity_baseline.json holds the float results of the HIP path on the MI355X as they are at the commit it names.
# The kernels are deterministic, so an unchanged build reproduces them exactly on any box; a change of a summation order
# moves them.  Nothing used to fail until the 1e-3 mm bar itself (Shelf went 3.1e-4 -> 4.3e-4 mm in round 5 unnoticed in the
# documents): now a fixture / sweep figure that GROWS past the tolerance below fails the GPU suite unless the baseline file
# is updated in the same commit (tools/update_parity_baseline.py) - the drift is then visible in the diff.
DRIFT_TOL = 1.10            # a tracked maximum may grow by 10 % before the baseline has to be re-pinned
DRIFT_FLOOR_FACTOR = 1.5    # conditioned Panoptic / Shelf fixtures: |build - ref32| max <= 1.5 x the reference's own fp32-vs-fp64 floor
DRIFT_FRAC_TOL = 0.003      # fraction of sweep joints within 1e-3 mm may drop by 0.3 points


def parity_baseline():
    import json
    with open(os.path.join(GOLDEN_DIR, "parity_baseline.json")) as f:
        return json.load(f)


def drift_check(kind, name, rep):
    """kind = 'fixtures' (rep of check_outputs) or 'sweeps' (summary of seed_sweep.replay).  Returns the findings; the
    callers assert that the list is empty."""
    if os.environ.get("FVP_TEST_DIAG_LIB") == "1":
        return []          # tools/gpu_switch_matrix.sh: alternative kernels (other summation orders) on purpose; the bars above still apply
    base = parity_baseline()[kind].get(name)
    if base is None:
        return [f"{name}: no entry in tests/golden/parity_baseline.json (run tools/update_parity_baseline.py)"]
    bad = []

    def grew(key, tol=DRIFT_TOL):
        if key in base and key in rep and rep[key] > tol * base[key] + 1e-12:
            bad.append(f"{name}: {key} {rep[key]:.4g} > {tol:.2f} x baseline {base[key]:.4g}")

    if kind == "fixtures":
        grew("max_mm_vs_fp64")
        if is_conditioned(name):
            grew("max_mm_vs_ref")
            if name not in FLOOR_RULE and rep["max_mm_vs_ref"] > DRIFT_FLOOR_FACTOR * rep["ref_floor_mm"] + 1e-12:
                bad.append(f"{name}: max_mm_vs_ref {rep['max_mm_vs_ref']:.3e} > {DRIFT_FLOOR_FACTOR} x the reference's own floor "
                           f"{rep['ref_floor_mm']:.3e}")
            for key in ("worst_ratio_vs_fp64", "worst_ratio_vs_ref32"):
                grew(key)
    else:
        for key in ("max_mm_r1q", "max_mm_where_floor_le_4e-4", "worst_err_over_proposal_floor", "worst_proposal_err_over_own_floor"):
            grew(key)
        if rep.get("violations_where_floor_le_4e-4", 0) > base.get("violations_where_floor_le_4e-4", 0):
            bad.append(f"{name}: literal-R1 violations {rep['violations_where_floor_le_4e-4']} > baseline {base['violations_where_floor_le_4e-4']}")
        if rep["frac_within_1e-3_mm"] < base["frac_within_1e-3_mm"] - DRIFT_FRAC_TOL:
            bad.append(f"{name}: fraction of joints within 1e-3 mm {rep['frac_within_1e-3_mm']:.4f} < baseline "
                       f"{base['frac_within_1e-3_mm']:.4f} - {DRIFT_FRAC_TOL}")
        if rep["joints"] != base["joints"] or rep["proposals"] != base["proposals"]:
            bad.append(f"{name}: compared {rep['joints']} joints / {rep['proposals']} proposals, baseline {base['joints']} / {base['proposals']}")
    return bad


def is_conditioned(case):
    from cases import CASES
    return CASES[case][1] == "c"


def load_golden(case):
    return np.load(os.path.join(GOLDEN_DIR, case + ".npz"))


def joint_errors(fused, g):
    v = g["valid"]
    f = fused[..., :3].detach().cpu().numpy()
    e32 = np.linalg.norm((f - g["fused_poses"][..., :3])[v], axis=-1)
    e64 = np.linalg.norm((f - g["floor_fused"])[v], axis=-1)
    return e32, e64, float(g["margins"][5])


def check_outputs(case, g, fused, planes, centers, engine, report=None):
    """Exact checks on integer / index work, tolerance checks on floating point."""
    v = g["valid"]
    last = engine.last
    # --- HDN: cubes are bit-exact by construction (same fp32 op order as the reference's CPU path).  The forward
    # materialises them only when engine.keep_hdn_cubes is set (the callers of this function run both ways).
    sx, sy = g["cubes_sub_stride"]
    if last["cubes"] is not None:
        cubes = last["cubes"].detach().cpu()
        assert np.array_equal(cubes[:, :, ::sx, ::sy, :].numpy(), g["cubes_sub"]), "HDN cubes differ"
        np.testing.assert_allclose(cubes.double().sum(dim=(1, 2, 3, 4)).numpy(), g["cubes_sum"], rtol=1e-12)
        np.testing.assert_allclose((cubes.double() ** 2).sum(dim=(1, 2, 3, 4)).numpy(), g["cubes_sq"], rtol=1e-12)
        assert torch.equal(last["zmax"].cpu(), cubes.max(dim=4)[0]), "fused z-max != max over the cubes"
        # the proposal columns (projected directly or gathered) are columns of these cubes
        Bc, Jc = cubes.shape[:2]
        cols = cubes.flatten(2, 3).permute(0, 2, 1, 3)                              # [B, XY, J, Z]
        fl = last["flat"].cpu()
        want = torch.stack([cols[b][fl[b]] for b in range(Bc)])                    # [B, N, J, Z]
        assert torch.equal(last["feat1d"].cpu().view(want.shape), want), "proposal z-columns != columns of the cubes"
    # --- HDN's public outputs (human_detection_net.py:76-104: hm2d, hm1d, proposal_centers, bbox_preds) against the
    # reference's: conv outputs in another summation order (same bar as the conv-stack tests), indices exact
    hm2d = last["hm2d"].detach().cpu().numpy()
    assert hm2d.shape[1] == 1
    np.testing.assert_allclose(hm2d[:, 0], g["hm2d"], rtol=1e-4, atol=2e-5, err_msg="hm2d (CenterNet heatmap head)")
    np.testing.assert_allclose(last["hm1d"].detach().cpu().numpy(), g["hm1d"], rtol=1e-4, atol=2e-5,
                               err_msg="hm1d (C2CNet output)")
    Bh, _, Xh, Yh = hm2d.shape
    bbox = last["bbox_flat"].detach().cpu().numpy()                                   # [B, X*Y, 2] (:88)
    assert bbox.shape == (Bh, Xh * Yh, 2)
    bmap = bbox.reshape(Bh, Xh, Yh, 2).transpose(0, 3, 1, 2)                          # back to [B,2,X,Y]
    np.testing.assert_allclose(bmap[:, :, ::sx, ::sy], g["bbox_map_sub"], rtol=1e-4, atol=2e-5,
                               err_msg="bbox_preds (CenterNet size head)")
    assert np.array_equal(bmap, last["bbox_map"].detach().cpu().numpy()), "bbox_preds is not the flattened size map"
    # topk_index [B,N,3] int64 (:98): (ix, iy) from the flat index with the reference's divisor quirk (get_index2D
    # divides by shape[1] = X), iz = the z arg-max that produced the bit-equal proposal centres
    ti = last["topk_index"].cpu().numpy()
    assert ti.dtype == np.int64
    assert np.array_equal(ti[..., 0], g["topk_flat"] // Xh) and np.array_equal(ti[..., 1], g["topk_flat"] % Xh)
    assert np.array_equal(ti[..., 2], np.argmax(g["hm1d"], axis=2)), "z arg-max index differs from the reference's hm1d"
    # --- proposals: indices exact
    assert np.array_equal(last["flat"].cpu().numpy(), g["topk_flat"]), "top-k flat indices differ"
    c = centers.detach().cpu().numpy()
    gc = g["proposal_centers"]
    # ABI 8: the mask of faster_voxelpose.py:45 comes out of fvp_proposals as uint8 flags
    if "valid" in last:
        assert last["valid"].dtype == torch.uint8 and np.array_equal(last["valid"].cpu().numpy().astype(bool), gc[..., 3] >= 0)
    assert np.array_equal(c[..., :3], gc[..., :3]), "proposal centres (mm) not bit-equal"
    assert np.array_equal(c[..., 3], gc[..., 3]), "valid flags differ"
    np.testing.assert_allclose(c[..., 5:7], gc[..., 5:7], rtol=0, atol=5e-5)
    np.testing.assert_allclose(c[..., 4], gc[..., 4], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(last["conf2d"].cpu().numpy(), g["conf2d"], rtol=1e-4, atol=1e-5)
    # --- JLN integer window arithmetic: exact
    N = v.shape[1]
    boxes = engine.last_jln["boxes"].cpu().numpy().reshape(v.shape[0], N, 9)
    offset = engine.last_jln["offset"].cpu().numpy().reshape(v.shape[0], N, 3)
    for f in range(v.shape[0]):
        if f"jl{f}_tl" not in g:
            continue
        assert np.array_equal(boxes[f][v[f]][:, 0:3], g[f"jl{f}_tl"])
        assert np.array_equal(boxes[f][v[f]][:, 3:6], g[f"jl{f}_start"])
        assert np.array_equal(boxes[f][v[f]][:, 6:9], g[f"jl{f}_end"])
        assert np.array_equal(offset[f][v[f]], g[f"jl{f}_offset"])
        # tri-planes: bit-exact (max of bit-exact samples)
        P = int(v[f].sum())
        planes_f = engine.last_jln["planes"].cpu().reshape(v.shape[0], N, 3, *engine.last_jln["planes"].shape[2:])[f][
            torch.from_numpy(v[f])]                                   # [P,3,J,C,C]
        tri = torch.cat([planes_f[:, 0], planes_f[:, 1], planes_f[:, 2]])   # reference order [3P,J,C,C]
        np.testing.assert_allclose(tri.double().sum(dim=(1, 2, 3)).numpy(), g[f"jl{f}_tri_sum"], rtol=1e-12)
        cs = int(g[f"jl{f}_tri_cstride"])
        rows = g[f"jl{f}_tri_rows"]
        assert np.array_equal(tri[rows][:, ::cs].numpy(), g[f"jl{f}_tri"]), "tri-plane maxima differ"
        # P2PNet features / WeightNet weights: fp32 conv in a different summation order
        feat = engine.last_jln["feat"].cpu().reshape(v.shape[0], N, 3, *engine.last_jln["feat"].shape[1:])[f][
            torch.from_numpy(v[f])]
        ft = torch.cat([feat[:, 0], feat[:, 1], feat[:, 2]])
        np.testing.assert_allclose(ft[rows][:, ::cs].numpy(), g[f"jl{f}_feat"], rtol=0, atol=2e-5)
        w = engine.last_jln["wgt"].cpu().reshape(v.shape[0], N, 3, -1)[f][torch.from_numpy(v[f])]
        wt = torch.cat([w[:, 0], w[:, 1], w[:, 2]])
        np.testing.assert_allclose(wt.numpy(), g[f"jl{f}_weights"], rtol=0, atol=2e-5)
    # --- final joints
    e32, e64, floor = joint_errors(fused, g)
    if report is not None:
        report.update(case=case, max_mm_vs_ref=float(e32.max()) if e32.size else 0.0,
                      mean_mm_vs_ref=float(e32.mean()) if e32.size else 0.0,
                      max_mm_vs_fp64=float(e64.max()) if e64.size else 0.0, ref_floor_mm=floor)
    if case in FLOOR_RULE and e32.size:
        floor_rule_check(case, fused[..., :3].detach().cpu().numpy(), g, report)
    elif e32.size:
        if is_conditioned(case):
            ok = e32.max() <= BAR_MM                       # the north-star bar itself
        else:
            ok = e32.max() <= BAR_MM or (e32.max() <= FLOOR_FACTOR * floor and
                                         e64.max() <= FP64_EXCEPTIONS.get(case, BAR_MM))
        assert ok, f"{case}: |build-ref32| max {e32.max():.2e} mm, |build-ref64| max {e64.max():.2e} mm, " \
                   f"reference fp32 noise floor {floor:.2e} mm"
    f = fused.detach().cpu().numpy()
    assert np.array_equal(f[..., 3], g["fused_poses"][..., 3])
    assert np.all(f[..., :3][~v] == 0.0), "invalid proposals must have zero joints"
    np.testing.assert_allclose(f[..., 4], g["fused_poses"][..., 4], rtol=2e-4, atol=1e-6)
    pl = planes.detach().cpu().numpy()
    tol = BAR_MM if (is_conditioned(case) and case not in FLOOR_RULE) else max(BAR_MM, FLOOR_FACTOR * floor) * 2
    np.testing.assert_allclose(pl, g["plane_poses"], rtol=0, atol=tol)


def run_custom_conv_stack(lib, device, spec, weights, x, stream=None, plane_valid=None):
    """Run a hand-made ``netspec.StackSpec`` (finalized) through fvp_pack_conv + fvp_conv_stack_run.
    ``weights``: {key + '.weight' / '.bias': tensor}; ``x``: [planes, C, H, W].  Returns every activation buffer."""
    import ctypes as C

    from faster_voxelpose_amd import _capi as capi
    blob = torch.zeros(max(spec.nparams, 4), device=device)
    s = stream
    for key, bn, transposed, oi in spec.param_keys:
        assert bn is None
        w = weights[key + ".weight"].to(device).contiguous()
        b = weights[key + ".bias"].to(device).contiguous()
        capi.check(lib, lib.fvp_pack_conv(C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, 1e-5,
                                          1 if transposed else 0, C.byref(spec.op_array[oi]), C.c_void_p(blob.data_ptr()), s),
                   "fvp_pack_conv")
    planes = x.shape[0]
    bufs = [x.to(device).contiguous()] + [torch.empty((planes,) + tuple(b), device=device) for b in spec.bufs[1:]]
    arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
    pv = None if plane_valid is None else plane_valid.to(device=device, dtype=torch.uint8).contiguous()
    capi.check(lib, lib.fvp_conv_stack_run(spec.op_array, len(spec.ops), C.c_void_p(blob.data_ptr()), arr, len(bufs), planes,
                                           None if pv is None else C.c_void_p(pv.data_ptr()), 1, s), "fvp_conv_stack_run")
    return bufs


def split_k_stack(cin, cmid, hw, seed=0):
    """conv3x3 cin->cmid (ReLU) then conv3x3 cmid->cmid + residual (ReLU) on a map that is NOT a power of two
    (so the direct kernel runs, and with >= 64 channels on <= 40x40 its split-K form): spec, weights, torch reference."""
    import torch.nn.functional as F

    from faster_voxelpose_amd import netspec
    spec = netspec.StackSpec(2, cin, hw)
    spec._conv_entries("a", cin, cmid, 3)
    spec._conv_entries("b", cmid, cmid, 3)
    h = spec.conv("a", None, 0, cmid, 3, relu=True)
    o = spec.conv("b", None, h, cmid, 3, relu=True, res=h)
    spec.outputs["out"] = o
    spec.finalize()
    g = torch.Generator().manual_seed(seed)
    w = {"a.weight": torch.randn(cmid, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, "a.bias": torch.randn(cmid, generator=g) * 0.1,
         "b.weight": torch.randn(cmid, cmid, 3, 3, generator=g) / (cmid * 9) ** 0.5, "b.bias": torch.randn(cmid, generator=g) * 0.1}

    def ref(x):
        x = x.double()
        h1 = F.relu(F.conv2d(x, w["a.weight"].double(), w["a.bias"].double(), padding=1))
        return F.relu(F.conv2d(h1, w["b.weight"].double(), w["b.bias"].double(), padding=1) + h1)
    return spec, w, ref, o


def front7_stack(cin, hw, seed=0):
    """The 7x7 front conv of P2PNet / CenterNet (cnns_2d.py:119-121: cin = joints -> 16, BN folded away here, ReLU) on a map
    whose width k_conv7 takes (64 / 80 / 128): spec, weights, float64 torch reference, output buffer id."""
    import torch.nn.functional as F

    from faster_voxelpose_amd import netspec
    spec = netspec.StackSpec(2, cin, hw)
    spec._conv_entries("f", cin, 16, 7)
    o = spec.conv("f", None, 0, 16, 7, relu=True)
    spec.outputs["out"] = o
    spec.finalize()
    g = torch.Generator().manual_seed(seed)
    w = {"f.weight": torch.randn(16, cin, 7, 7, generator=g) / (cin * 49) ** 0.5, "f.bias": torch.randn(16, generator=g) * 0.1}

    def ref(x):
        return F.relu(F.conv2d(x.double(), w["f.weight"].double(), w["f.bias"].double(), padding=3))
    return spec, w, ref, o


def reg_stack(seed=0, fused_head=True, head_cout=15):
    """1x1 convs and transposed convs with P2PNet's channel counts on 16x16 / 8x8 maps (whole 32-pixel tiles): the layers
    k_conv_reg takes.  x [planes, 32, 16, 16] -> 1x1 32->64, pool, 1x1 64->128, up 128->64 (+ skip), up 64->32 (+ skip),
    1x1 32->15 (fused into the second transposed conv when it is that conv's only consumer), 1x1 16->32 on a slice-free
    side branch.  Returns spec, weights, float64 torch reference (dict of outputs), output buffer ids."""
    import torch.nn.functional as F

    from faster_voxelpose_amd import netspec
    spec = netspec.StackSpec(2, 32, (16, 16))
    spec._conv_entries("s64", 32, 64, 1)
    spec._conv_entries("t128", 64, 128, 1)
    spec._conv_entries("u64", 128, 64, 2, transposed=True)
    spec._conv_entries("u32", 64, 32, 2, transposed=True)
    spec._conv_entries("head", 32, head_cout, 1)
    spec._conv_entries("d16", 32, 16, 3)
    spec._conv_entries("s32", 16, 32, 1)
    s64 = spec.conv("s64", None, 0, 64, 1, relu=False)
    p = spec.pool(s64)
    t128 = spec.conv("t128", None, p, 128, 1, relu=True)
    u64 = spec.up("u64", None, t128, 64, s64)
    u32 = spec.up("u32", None, p, 32, 0)
    head = spec.conv("head", None, u32, head_cout, 1, relu=False)
    d16 = spec.conv("d16", None, u32 if not fused_head else 0, 16, 3, relu=True)
    s32 = spec.conv("s32", None, d16, 32, 1, relu=False, res=0)
    spec.outputs.update(u64=u64, head=head, s32=s32)
    spec.finalize()
    g = torch.Generator().manual_seed(seed)

    def wb(key, cout, cin, k, transposed=False):
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        return {key + ".weight": torch.randn(shape, generator=g) / (cin * k * k) ** 0.5, key + ".bias": torch.randn(cout, generator=g) * 0.1}
    w = {}
    for args in (("s64", 64, 32, 1), ("t128", 128, 64, 1), ("u64", 64, 128, 2, True), ("u32", 32, 64, 2, True),
                 ("head", head_cout, 32, 1), ("d16", 16, 32, 3), ("s32", 32, 16, 1)):
        w.update(wb(*args))

    def ref(x):
        x = x.double()
        W = {k: v.double() for k, v in w.items()}
        a = F.conv2d(x, W["s64.weight"], W["s64.bias"])
        pp = F.max_pool2d(a, 2)
        t = F.relu(F.conv2d(pp, W["t128.weight"], W["t128.bias"]))
        o_u64 = F.relu(F.conv_transpose2d(t, W["u64.weight"], W["u64.bias"], stride=2)) + a
        o_u32 = F.relu(F.conv_transpose2d(pp, W["u32.weight"], W["u32.bias"], stride=2)) + x
        o_head = F.conv2d(o_u32, W["head.weight"], W["head.bias"])
        d = F.relu(F.conv2d(o_u32 if not fused_head else x, W["d16.weight"], W["d16.bias"], padding=1))
        o_s32 = F.conv2d(d, W["s32.weight"], W["s32.bias"]) + x
        return dict(u64=o_u64, head=o_head, s32=o_s32)
    return spec, w, ref, dict(u64=u64, head=head, s32=s32)
