"""Synthetic configs, cameras, heatmaps and weights for tests / smoke / bench.

Measurement / test infrastructure: lives beside ``bench.py`` at the repo root, outside the product
package (which never imports it).

No dataset or checkpoint can be fetched offline, so every measurement and parity case in
this repo runs on the recipes below.  Everything is derived from
numpy ``PCG64`` streams keyed by (seed, name), so the build container (which generates
the golden vectors from the reference) and the GPU box reproduce identical inputs.

Shape sets follow the reference's shipped YAMLs:
``configs/panoptic/jln64.yaml``, ``configs/shelf/jln64.yaml``, ``configs/campus/jln64.yaml``;
camera files are the reference's in-repo calibrations (data fixtures under
``tests/golden/``).  List-valued fields stay Python lists (as a YAML overlay delivers
them) so that model constants come out fp32 / int like the reference's
("type trap").
"""
import json
import os
import zlib
from types import SimpleNamespace as NS

import numpy as np
import torch

_FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")

SHAPES = {
    "panoptic": dict(V=5, J=15, hm=[240, 128], img=[960, 512], ori=[1920, 1080],
                     space=[8000.0, 8000.0, 2000.0], center=[0.0, -500.0, 800.0],
                     voxels=[80, 80, 20], N=10, min_score=0.3,
                     calib="calibration_panoptic_demo.json", seq="customized_sequence"),
    "shelf": dict(V=5, J=17, hm=[200, 152], img=[800, 608], ori=[1032, 776],
                  space=[8000.0, 8000.0, 2000.0], center=[450.0, -320.0, 800.0],
                  voxels=[80, 80, 20], N=10, min_score=0.1,
                  calib="calibration_shelf.json", seq="shelf"),
    "campus": dict(V=3, J=17, hm=[200, 160], img=[800, 640], ori=[360, 288],
                   space=[12000.0, 12000.0, 2000.0], center=[3000.0, 4500.0, 1000.0],
                   voxels=[80, 80, 20], N=5, min_score=0.1,
                   calib="calibration_campus.json", seq="campus"),
    # BASELINE.json configs[3]: the Panoptic cameras with a 128x128x32 detection grid and jln128
    "panoptic128": dict(V=5, J=15, hm=[240, 128], img=[960, 512], ori=[1920, 1080],
                        space=[8000.0, 8000.0, 2000.0], center=[0.0, -500.0, 800.0],
                        voxels=[128, 128, 32], N=10, min_score=0.3, cube=[128, 128, 128],
                        calib="calibration_panoptic_demo.json", seq="customized_sequence"),
    # not a reference config: a miniature (Campus cameras) for fast kernel-logic tests
    "tiny": dict(V=3, J=5, hm=[50, 40], img=[200, 160], ori=[360, 288],
                 space=[12000.0, 12000.0, 2000.0], center=[3000.0, 4500.0, 1000.0],
                 voxels=[16, 16, 8], N=3, min_score=0.1, cube=[16, 16, 16],
                 calib="calibration_campus.json", seq="campus"),
}


def make_cfg(name="panoptic", device="cpu", min_score=None, voxels=None, cube=None, max_people=None):
    """Attribute-style cfg with exactly the fields the hot path reads."""
    s = SHAPES[name]
    return NS(
        NAME=name, DEVICE=device,
        DATASET=NS(IMAGE_SIZE=list(s["img"]), HEATMAP_SIZE=list(s["hm"]),
                   ORI_IMAGE_SIZE=list(s["ori"]), NUM_JOINTS=s["J"], CAMERA_NUM=s["V"]),
        CAPTURE_SPEC=NS(SPACE_SIZE=list(s["space"]), SPACE_CENTER=list(s["center"]),
                        VOXELS_PER_AXIS=list(voxels or s["voxels"]),
                        MAX_PEOPLE=int(max_people or s["N"]),
                        MIN_SCORE=s["min_score"] if min_score is None else min_score),
        INDIVIDUAL_SPEC=NS(SPACE_SIZE=[2000.0, 2000.0, 2000.0],
                           VOXELS_PER_AXIS=list(cube or s.get("cube", [64, 64, 64]))),
        NETWORK=NS(BETA=100, NUM_CHANNEL_JOINT_FEAT=32, NUM_CHANNEL_JOINT_HIDDEN=64),
        TRAIN=NS(LAMBDA_LOSS_2D=1.0, LAMBDA_LOSS_1D=1.0, LAMBDA_LOSS_BBOX=0.1, LAMBDA_LOSS_FUSED=5.0),
    )


def load_cameras(name):
    """``{seq: [cam dict, ...]}`` in the convention the reference's forward takes:
    Panoptic demo = list of dicts of nested lists (demo/visualize.ipynb cell 9);
    Shelf/Campus = int-keyed dict of numpy arrays (lib/dataset/shelf.py:143-152)."""
    s = SHAPES[name]
    with open(os.path.join(_FIXTURES, s["calib"])) as f:
        raw = json.load(f)
    if name.startswith("panoptic"):
        return {s["seq"]: raw[s["seq"]]}, s["seq"]
    if name == "tiny":
        return {s["seq"]: [raw[k] for k in sorted(raw)]}, s["seq"]
    cams = {int(k): {kk: np.array(vv) for kk, vv in cam.items()} for k, cam in raw.items()}
    return {s["seq"]: cams}, s["seq"]


def resize_transform(cfg):
    from faster_voxelpose_amd.utils.transforms import get_resize_transform
    return torch.as_tensor(get_resize_transform(cfg.DATASET.ORI_IMAGE_SIZE, cfg.DATASET.IMAGE_SIZE),
                           dtype=torch.float32)


def _rng(seed, name=""):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def heatmaps_uniform(cfg, batch, seed=2):
    """Flavour (U): iid uniform [0,1) heatmaps ``[B,V,J,H,W]``."""
    w, h = cfg.DATASET.HEATMAP_SIZE
    shape = (batch, cfg.DATASET.CAMERA_NUM, cfg.DATASET.NUM_JOINTS, h, w)
    return torch.from_numpy(_rng(seed, "heat_u").random(shape, dtype=np.float32))


def _project_np(pts, cam):
    """float64 pinhole + distortion, only used to place synthetic blobs."""
    R = np.asarray(cam["R"], np.float64).reshape(3, 3)
    T = np.asarray(cam["T"], np.float64).reshape(3, 1)
    k = np.asarray(cam["k"], np.float64).reshape(3)
    p = np.asarray(cam["p"], np.float64).reshape(2)
    xc = R @ (pts.T - T)
    y = xc[:2] / (xc[2] + 1e-5)
    r = (y ** 2).sum(0)
    d = 1 + k[0] * r + k[1] * r * r + k[2] * r * r * r
    u = y[0] * d + 2 * p[0] * y[0] * y[1] + p[1] * (r + 2 * y[0] * y[0])
    v = y[1] * d + 2 * p[1] * y[0] * y[1] + p[0] * (r + 2 * y[1] * y[1])
    return np.stack([cam["fx"] * u + cam["cx"], cam["fy"] * v + cam["cy"]], 1), xc[2]


def heatmaps_blobs(cfg, cameras, seq, batch, people=4, seed=3, sigma=3.0):
    """Flavour (G): Gaussian blobs (sigma px in heatmap space) at the projections of
    ``people`` synthetic skeletons per frame, max-combined per joint, clipped to [0,1].
    Mirrors what the reference's data layer feeds on the precomputed-heatmap path
    (lib/dataset/JointsDataset.py:271-338) without sharing its code."""
    from faster_voxelpose_amd.utils.transforms import get_resize_transform
    w, h = cfg.DATASET.HEATMAP_SIZE
    V, J = cfg.DATASET.CAMERA_NUM, cfg.DATASET.NUM_JOINTS
    rt = get_resize_transform(cfg.DATASET.ORI_IMAGE_SIZE, cfg.DATASET.IMAGE_SIZE)
    feat = np.array([w, h], np.float64) / np.array(cfg.DATASET.IMAGE_SIZE, np.float64)
    cams = cameras[seq]
    cams = [cams[i] for i in range(len(cams))]
    size = np.array(cfg.CAPTURE_SPEC.SPACE_SIZE)
    cen = np.array(cfg.CAPTURE_SPEC.SPACE_CENTER)
    rng = _rng(seed, "heat_g")
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((batch, V, J, h, w), np.float32)
    for b in range(batch):
        roots = np.stack([rng.uniform(cen[0] - 0.35 * size[0], cen[0] + 0.35 * size[0], people),
                          rng.uniform(cen[1] - 0.35 * size[1], cen[1] + 0.35 * size[1], people),
                          np.full(people, 900.0)], 1)
        for k in range(people):
            joints = roots[k] + rng.normal(0.0, 1.0, (J, 3)) * np.array([180.0, 180.0, 400.0])
            for v in range(V):
                px, depth = _project_np(joints, cams[v])
                px = (rt[:, :2] @ px.T + rt[:, 2:3]).T * feat
                for j in range(J):
                    if depth[j] <= 0 or not (-3 * sigma <= px[j, 0] < w + 3 * sigma) \
                            or not (-3 * sigma <= px[j, 1] < h + 3 * sigma):
                        continue
                    g = np.exp(-((xs - px[j, 0]) ** 2 + (ys - px[j, 1]) ** 2) / (2 * sigma ** 2))
                    out[b, v, j] = np.maximum(out[b, v, j], g.astype(np.float32))
    return torch.from_numpy(np.clip(out, 0.0, 1.0))


# ---- synthetic weights -----------------------------------------------------------------
# Head gains fixed once (calibrated in the build container on blob heatmaps) so that the
# soft-argmax is moderately peaked, proposals fire, and bbox sizes vary around 0.6:
HEAD_GAIN = {
    "joint_net.conv_net.output_layer.weight": 0.005,
    "pose_net.center_net.output_hm.2.weight": 0.25,
    "pose_net.c2c_net.output_hm.weight": 0.25,
    "pose_net.center_net.output_size.2.weight": 0.02,
}
HEAD_BIAS = {
    "pose_net.center_net.output_hm.2.bias": 0.6,
    "pose_net.c2c_net.output_hm.bias": 0.9,
    "pose_net.center_net.output_size.2.bias": 0.6,
}


def fill_state_dict(state_dict, seed=7, bbox_bias=None):
    """Seeded, well-conditioned weights for every key of a (reference-compatible)
    state_dict: conv / linear ~ N(0, 2/fan_in) (variance-preserving through ReLU), zero
    conv bias, BatchNorm with mildly randomised running stats so the BN arithmetic is
    actually exercised.  Returns a new dict of fp32 CPU tensors (int64 for
    ``num_batches_tracked``)."""
    out = {}
    for key, ref in state_dict.items():
        shape = tuple(ref.shape)
        rng = _rng(seed, key)
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        is_bn = (key.rsplit(".", 1)[0] + ".running_mean") in state_dict
        if is_bn:
            if leaf == "weight":
                a = rng.uniform(0.5, 1.5, shape)
            elif leaf == "bias":
                a = rng.normal(0.0, 0.1, shape)
            elif leaf == "running_mean":
                a = rng.normal(0.0, 0.1, shape)
            else:
                a = rng.uniform(0.5, 1.5, shape)
        elif leaf == "weight":
            transposed = "upsample" in key
            fan_in = (shape[0] if transposed else int(np.prod(shape[1:])))
            a = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape) * HEAD_GAIN.get(key, 1.0)
        else:
            a = np.full(shape, HEAD_BIAS.get(key, 0.0))
            if key == "pose_net.center_net.output_size.2.bias" and bbox_bias is not None:
                a = np.full(shape, bbox_bias)
        out[key] = torch.from_numpy(np.asarray(a, np.float32))
    return out


def fill_backbone_state_dict(state_dict, seed=7):
    """Seeded weights for a Pose-ResNet state_dict that keep activations O(1) through the 16
    residual blocks in eval mode (the last BatchNorm of every block is damped, transposed convs are
    scaled for their effective fan-in of Cin * 4)."""
    out = fill_state_dict(state_dict, seed=seed)
    for k in out:
        if k.startswith("layer") and (k.endswith(".bn3.weight") or (k.endswith(".bn2.weight") and (k[:-10] + "bn3.weight") not in out)):
            out[k] = out[k] * 0.3
        if k.startswith("deconv_layers") and k.endswith(".weight") and out[k].dim() == 4:
            cin, cout = out[k].shape[:2]
            out[k] = out[k] * float(np.sqrt((cout * 16) / (cin * 4.0)))
        if k == "final_layer.bias":
            out[k] = torch.full_like(out[k], 0.05)
    return out


# ---- conditioned weights (float-parity fixtures) ------------------------------------------
def _pass_through(w, transposed, res_branch):
    """Add the channel-preserving identity-like kernel to a conv weight ``w`` (numpy, in place):
    conv [cout, cin, *k] / transposed conv [cin, cout, *k].  Where the width changes, channel c
    feeds channel c mod cout (narrowing, the wrapped-around channels at half weight so that no two
    inputs tie) or is fed by channel c mod cin (widening)."""
    k = w.shape[2:]
    centre = tuple(n // 2 for n in k)
    if transposed:                                   # k2 s2: every tap, pairs of inputs fold into one output
        cin, cout = w.shape[:2]
        for ci in range(cin):
            w[(ci, ci % cout) + (slice(None),) * len(k)] += 0.5
        return
    cout, cin = w.shape[:2]
    amp = 0.25 if res_branch else 1.0
    if cin >= cout:
        for ci in range(cin):
            w[(ci % cout, ci) + centre] += amp if ci < cout else 0.5 * amp
    else:
        for co in range(cout):
            w[(co, co % cin) + centre] += amp


def fill_state_dict_conditioned(state_dict, seed=7, texture=0.1, jln_gain=0.25, hm2d_gain=0.5, hm1d_gain=0.5,
                                hdn_cut=0.0):
    """Weights for the float-parity fixtures (SURVEY.md section 7 hard part 1 / section 8d).

    The nets are built so that they behave like trained ones in the respect that matters for a
    1e-3 mm comparison: the detection heat maps peak at the people, and every joint map that feeds
    the soft-argmax (beta = 100) has ONE clearly dominant, moderately peaked mode.  Purely random
    weights (``fill_state_dict``) detect nothing meaningful and give multi-modal joint maps with
    near-ties, for which the reference's own fp32 result is only reproducible to ~1e-2 mm.

    Construction: every conv is ``pass-through + texture``: a channel-preserving identity-like
    kernel (centre tap) plus ``texture`` x the generic random kernel; every BatchNorm has unit gain
    up to +-10 % (gamma = sqrt(var) * U(0.9, 1.1); the closing BatchNorm of a residual branch at half
    gain) with small random running mean / beta.  All weights stay non-zero and sign-mixed, so every
    multiply-add of the kernels is exercised.  The heads read the channels that carry 'their' joint (joint 0 for the two detection heads); their gains
    set the confidence scale (detection) and the peakedness of the soft-argmax (joint net).  The
    bounding-box head and WeightNet keep the generic recipe."""
    out = fill_state_dict(state_dict, seed=seed)
    head_gain = {"joint_net.conv_net.output_layer.weight": jln_gain,
                 "pose_net.center_net.output_hm.2.weight": hm2d_gain,
                 "pose_net.c2c_net.output_hm.weight": hm1d_gain}
    for key in out:
        if not key.startswith(("joint_net.conv_net.", "pose_net.center_net.", "pose_net.c2c_net.")):
            continue
        if ".output_size." in key:
            continue
        rng = _rng(seed, key + "/conditioned")
        stem, leaf = key.rsplit(".", 1)
        is_bn = (stem + ".running_mean") in out
        shape = tuple(out[key].shape)
        if is_bn:
            if leaf == "running_var":
                continue                                            # generic U(0.5, 1.5)
            if leaf == "weight":
                var = out[stem + ".running_var"].numpy().astype(np.float64)
                a = np.sqrt(var + 1e-5) * rng.uniform(0.9, 1.1, shape)
                if stem.endswith("res_branch.4"):
                    a = a * 0.5
            elif leaf == "bias":
                a = rng.normal(0.0, 0.02, shape)
                if stem == "pose_net.center_net.front_layers.0.block.1":
                    a = a - hdn_cut                                 # ReLU cut-off: drops weak multi-view ray crossings
            else:                                                   # running_mean
                a = rng.normal(0.0, 0.05, shape)
            out[key] = torch.from_numpy(np.asarray(a, np.float32))
            continue
        if leaf == "bias":
            if key in HEAD_BIAS:
                out[key] = torch.zeros(shape)
            continue
        if out[key].dim() < 3:
            continue
        w = out[key].numpy().astype(np.float64) / HEAD_GAIN.get(key, 1.0) * texture
        if key == "joint_net.conv_net.output_layer.weight":         # joint j <- channels j, j + 16 (wraps for J > 16)
            half = shape[1] // 2
            for j in range(shape[0]):
                w[j, j % half] += 0.5 if j < half else 0.25
                w[j, half + j % half] += 0.5 if j < half else 0.25
            w = w * head_gain[key]
        elif key in head_gain:                                      # detection heads: the channels carrying joint 0
            half = shape[1] // 2
            w[0, 0] += 0.5
            w[0, half] += 0.5
            w = w * head_gain[key]
        else:
            _pass_through(w, "upsample" in key, "res_branch" in key)
        out[key] = torch.from_numpy(np.asarray(w, np.float32))
    return out


def _place_people(cfg, rng, count, region, spacing, joint_std, root_z):
    """``count`` skeletons [J,3] (mm) on distinct cells of a ``spacing`` grid inside +-``region`` of the
    capture-space centre, roots jittered by 10 % of the spacing, joints within 2 sigma of the root."""
    J = cfg.DATASET.NUM_JOINTS
    cen = np.array(cfg.CAPTURE_SPEC.SPACE_CENTER)
    n = int(np.floor(2 * region / spacing)) + 1
    cells = np.array([(i, j) for i in range(n) for j in range(n)], np.float64) * spacing - region
    pick = rng.permutation(len(cells))[:count]
    people = []
    for c in cells[pick]:
        root = np.array([cen[0] + c[0], cen[1] + c[1], root_z]) + np.append(rng.uniform(-0.1, 0.1, 2) * spacing, 0.0)
        people.append(root + np.clip(rng.normal(0.0, 1.0, (J, 3)), -2.0, 2.0) * np.array(joint_std))
    return people


def heatmaps_people(cfg, cameras, seq, batch, people, seed=3, sigma=3.0, region=1200.0, spacing=1000.0,
                    joint_std=(150.0, 150.0, 250.0), root_z=900.0):
    """Flavour (C), for the float-parity fixtures: like ``heatmaps_blobs`` but the skeletons stand
    on distinct cells of a ``spacing``-mm grid inside +-``region`` mm of the capture-space centre
    (jittered by 10 % of the spacing), so that every person is seen by the cameras, no two people
    share a 2 m joint cube window, and coordinates stay small (fp32 ulp of the outputs).  ``people``
    may be an int or a per-frame list."""
    from faster_voxelpose_amd.utils.transforms import get_resize_transform
    w, h = cfg.DATASET.HEATMAP_SIZE
    V, J = cfg.DATASET.CAMERA_NUM, cfg.DATASET.NUM_JOINTS
    rt = get_resize_transform(cfg.DATASET.ORI_IMAGE_SIZE, cfg.DATASET.IMAGE_SIZE)
    feat = np.array([w, h], np.float64) / np.array(cfg.DATASET.IMAGE_SIZE, np.float64)
    cams = cameras[seq]
    cams = [cams[i] for i in range(len(cams))]
    rng = _rng(seed, "heat_c")
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((batch, V, J, h, w), np.float32)
    counts = [people] * batch if np.isscalar(people) else list(people)
    for b in range(batch):
        for joints in _place_people(cfg, rng, counts[b], region, spacing, joint_std, root_z):
            for v in range(V):
                px, depth = _project_np(joints, cams[v])
                px = (rt[:, :2] @ px.T + rt[:, 2:3]).T * feat
                for j in range(J):
                    if depth[j] <= 0:
                        continue
                    g = np.exp(-((xs - px[j, 0]) ** 2 + (ys - px[j, 1]) ** 2) / (2 * sigma ** 2))
                    out[b, v, j] = np.maximum(out[b, v, j], g.astype(np.float32))
    return torch.from_numpy(np.clip(out, 0.0, 1.0))


def pred2d_people(cfg, cameras, seq, people, seed=3, region=1200.0, spacing=1000.0,
                  joint_std=(150.0, 150.0, 250.0), root_z=900.0):
    """2-D detections of one frame for the precomputed-heatmap path (``db_rec['pred_pose2d']``
    convention: list over views of lists of [J,3] arrays = x, y in ORIGINAL image pixels, score):
    the projections of ``people`` consistent 3-D skeletons placed like ``heatmaps_people``."""
    cams = cameras[seq]
    cams = [cams[i] for i in range(len(cams))]
    rng = _rng(seed, "pred2d")
    skeletons = _place_people(cfg, rng, people, region, spacing, joint_std, root_z)
    frame = []
    for cam in cams:
        preds = []
        for joints in skeletons:
            px, depth = _project_np(joints, cam)
            preds.append(np.concatenate([px, np.ones((len(px), 1))], axis=1))
        frame.append(preds)
    return frame
