"""Edge-case checks shared by the emulator suite (CPU) and the GPU suite: every function takes the
loaded C-ABI library (``lib``: libfvp_hip.so bound by _capi on the GPU box, the emulated build on the
CPU) and a device string, and goes through the same ctypes / host code as the product."""
import ctypes as C
import os

import numpy as np
import torch

import fvp_oracle as O
import fvp_synthetic as S
from faster_voxelpose_amd import _capi as capi
from faster_voxelpose_amd.models import faster_voxelpose as FV

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(shape, lib, dev, **kw):
    cfg = S.make_cfg(shape, device=dev, **kw)
    m = FV.FasterVoxelPoseNet(cfg, _lib=lib) if lib is not None else FV.get(cfg).to(dev)
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=7))
    return m, cfg


def zero_batch_through_every_export(lib, dev):
    """B = 0 / n = 0 is a valid call of every batched export: returns success without touching memory."""
    m, cfg = _model("tiny", lib, dev)
    e = m.engine
    buf = torch.zeros(4096, device=dev)
    ibuf = torch.zeros(64, dtype=torch.int64, device=dev)
    p, ip = C.c_void_p(buf.data_ptr()), C.c_void_p(ibuf.data_ptr())
    g = e.geom(S.resize_transform(cfg).to(dev))
    g.V = cfg.DATASET.CAMERA_NUM
    s = e.stream()
    J, X, Y, Z, N, Cn = e.J, e.X, e.Y, e.Z, e.N, e.C
    L = e.lib
    calls = {
        "fvp_heatmaps_to_cl": lambda: L.fvp_heatmaps_to_cl(p, p, 0, C.byref(g), s),
        "fvp_project_whole": lambda: L.fvp_project_whole(p, p, ip, p, p, p, X, Y, Z, 0, C.byref(g), p, p, s),
        "fvp_project_columns": lambda: L.fvp_project_columns(p, p, ip, p, p, p, X, Y, Z, 0, C.byref(g), ip, N, p, s),
        "fvp_zmax": lambda: L.fvp_zmax(p, p, 0, Z, s),
        "fvp_person_boxes": lambda: L.fvp_person_boxes(p, 0, p, e.fine_cube, ip, p, s),
        "fvp_project_individual": lambda: L.fvp_project_individual(p, p, ip, ip, None, ip, p, p, p, ip, Cn, 0, C.byref(g), p, s),
        "fvp_triplane_max": lambda: L.fvp_triplane_max(p, p, 0, J, Cn, s),
        "fvp_project_individual_triplane": lambda: L.fvp_project_individual_triplane(p, p, ip, ip, None, ip, p, p, p, ip, Cn, 0,
                                                                                    C.byref(g), p, N, None, s),
        "fvp_nms_topk": lambda: L.fvp_nms_topk(p, 0, X, Y, N, p, ip, ip, s),
        "fvp_gather_proposals": lambda: L.fvp_gather_proposals(p, p, ip, 0, J, X, Y, Z, N, p, p, p, s),
        "fvp_proposals": lambda: L.fvp_proposals(p, p, ip, p, p, 0.1, 0, N, Z, ip, p, None, s),
        "fvp_softargmax_weightnet": lambda: L.fvp_softargmax_weightnet(p, p, p, 100.0, 0, J, Cn, e.F, e.Hd, None, p, p, p, s),
        "fvp_fuse_poses": lambda: L.fvp_fuse_poses(p, p, p, p, None, 0, J, p, p, p, s),
        "fvp_rasterise_heatmaps": lambda: L.fvp_rasterise_heatmaps(p, ip, 0, 2, J, e.W, e.H, 4.0, 4.0, 3.0, p, None, e.JP, s),
    }
    spec = e.specs["center_net"]
    arr = (C.c_void_p * len(spec.bufs))(*[buf.data_ptr()] * len(spec.bufs))
    calls["fvp_conv_stack_run"] = lambda: L.fvp_conv_stack_run(spec.op_array, len(spec.ops), C.c_void_p(e.params["center_net"].data_ptr()),
                                                               arr, len(spec.bufs), 0, None, 1, s)
    s1 = e.specs["c2c_net"]
    calls["fvp_conv_stack_run_fused_1d"] = lambda: L.fvp_conv_stack_run_fused_1d(s1.op_array, len(s1.ops),
                                                                                 C.c_void_p(e.params["c2c_net"].data_ptr()), p, p, 0, s)
    for name, fn in calls.items():
        assert fn() == 0, f"{name} rejects an empty batch"
    if dev != "cpu":
        torch.cuda.synchronize()
    assert float(buf.abs().sum()) == 0.0 and int(ibuf.abs().sum()) == 0, "an empty-batch call wrote to memory"
    # and through the module API: an empty proposal list
    pl = m.joint_net.project_layer
    heat = torch.zeros(1, cfg.DATASET.CAMERA_NUM, J, e.H, e.W, device=dev)
    cams, seq = S.load_cameras("tiny")
    cubes, offset = pl(heat, 0, {"seq": [seq]}, torch.zeros(0, 7, device=dev), cams, S.resize_transform(cfg).to(dev))
    assert cubes.shape == (0, J, Cn, Cn, Cn) and offset.shape == (0, 3)


def negative_bbox_gives_an_empty_window(lib, dev, shape="tiny"):
    """project_individual.py:116-126: a bounding-box estimate below 1 - 2 gives margin > C/2, i.e.
    start >= end: the person's cube stays all zero but the person is still processed."""
    m, cfg = _model(shape, lib, dev, min_score=-1.0)
    cams, seq = S.load_cameras(shape)
    rt = S.resize_transform(cfg)
    heat = S.heatmaps_uniform(cfg, 1, seed=3)
    meta = {"seq": [seq]}
    J, Cn = m.engine.J, m.engine.C
    cen = torch.tensor(cfg.CAPTURE_SPEC.SPACE_CENTER)
    pc = torch.zeros(3, 7)
    pc[:, :3] = cen
    pc[:, 5:7] = torch.tensor([[0.6, 0.6], [-0.5, 0.7], [0.7, -1.5]])      # person 1: x window empty, 2: y window empty
    with torch.no_grad():
        cubes, offset = m.joint_net.project_layer(heat.to(dev), 0, meta, pc.to(dev), cams, rt.to(dev))
    spec = O.IndividualSpec(cfg)
    ocubes, ooff, (tl, start, end) = O.project_individual(spec, cfg, heat[0], pc, [cams[seq][i] for i in range(len(cams[seq]))], rt)
    assert bool((start[1] >= end[1]).any()) and bool((start[2] >= end[2]).any()) and bool((start[0] < end[0]).all())
    boxes = m.joint_net.project_layer.last_boxes.cpu()
    assert torch.equal(boxes[:, 0:3], tl) and torch.equal(boxes[:, 3:6], start) and torch.equal(boxes[:, 6:9], end)
    cubes = cubes.cpu()
    assert float(cubes[1].abs().max()) == 0.0 and float(cubes[2].abs().max()) == 0.0
    # (the oracle's own bilinear restatement is within 2 ulp of 1.0 of F.grid_sample, the kernel is bit-equal to it)
    diff = (cubes[0] - ocubes[0]).abs()
    assert float(diff.max()) <= 2.5e-7 and float(cubes[0].max()) > 0, \
        (float(diff.max()), int((diff > 2.5e-7).sum()), float(cubes[0].max()), float(ocubes[0].max()),
         float(cubes[0].sum()), float(ocubes[0].sum()), tl.tolist(), start.tolist(), end.tolist())
    assert torch.equal(offset.cpu(), ooff)
    # fused fast path: the three planes of the empty-window people are zero, person 0 equals the max-projections
    pcs = torch.zeros(1, m.engine.N, 7)
    pcs[0, :, 3] = -1
    pcs[0, :3] = pc
    mask = torch.tensor([[True, True, True] + [False] * (m.engine.N - 3)], device=dev)
    with torch.no_grad():
        fused, planes = m.joint_net(meta, heat.to(dev), pcs.to(dev), mask, cams, rt.to(dev))
    tri = m.engine.last_jln["planes"].cpu()                      # [N,3,J,C,C]
    assert float(tri[1].abs().max()) == 0.0 and float(tri[2].abs().max()) == 0.0
    want = O.triplane_max(cubes[:1])
    assert torch.equal(tri[0, 0], want[0]) and torch.equal(tri[0, 1], want[1]) and torch.equal(tri[0, 2], want[2])
    assert bool(torch.isfinite(fused).all()) and bool(torch.isfinite(planes).all())


def sampling_grids_equal_reference(lib, dev, shape):
    """fvp_sample_grid on the whole-space grid and on the fine grid of the joint stage vs digests of the
    reference's cached grids (tests/golden/grids.npz): bit-equal."""
    g = np.load(os.path.join(GOLDEN, "grids.npz"))
    m, cfg = _model(shape, lib, dev)
    cams, seq = S.load_cameras(shape)
    rt = S.resize_transform(cfg).to(dev)
    e = m.engine
    V = cfg.DATASET.CAMERA_NUM
    e.frame_sets({"seq": [seq]}, cams, V)
    whole = e.sample_grid(e.whole_axes, seq, rt, V).cpu().squeeze(1)                        # [V, n, 2]
    ws = int(g[f"{shape}_whole_stride"])
    assert np.array_equal(whole[:, ::ws].numpy(), g[f"{shape}_whole"]), "whole-space sampling grid differs"
    fine = e.sample_grid(e.fine_axes, seq, rt, V).cpu().view(V, *e.fine, 2)
    assert list(g[f"{shape}_fine_dims"]) == list(e.fine)
    sx, sy, sz = (int(v) for v in g["fine_stride"])
    assert np.array_equal(fine[:, ::sx, ::sy, ::sz].numpy(), g[f"{shape}_fine"]), "fine sampling grid differs"
