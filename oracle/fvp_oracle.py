"""CPU oracle for the Faster-VoxelPose inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain restatement (torch-CPU / numpy, fp32 unless said otherwise) of what
the reference computes on the path  heatmaps -> project_layer -> HDN -> JLN -> 3D joints.
It is the checker for the HIP kernels: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``faster-voxelpose_amd/``) never imports anything from ``oracle/`` and has no CPU fallback.

Pinning: the reference ships no tests / golden vectors for this path,
so the oracle is pinned against outputs of the reference itself, produced in the build
container by ``tests/golden/make_golden.py`` (which imports ``/root/reference/lib``) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them.

All ``file:line`` citations are relative to the reference checkout (``/root/reference``).
The oracle works on a *flat state_dict* (the reference's checkpoint format,
lib/utils/utils.py:89-98) and a plain attribute-style cfg; it shares no code with the
product package.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------
# geometry (one-time per sequence)
# --------------------------------------------------------------------------------------
def axis_coords(size, center, nbins):
    """Voxel-centre coordinates along one axis: ``linspace(-S/2, S/2, n) + c``.
    lib/models/project_whole.py:34-40 (same code project_individual.py:50-56)."""
    return torch.linspace(-size / 2, size / 2, int(nbins)) + center


def compute_grid(space_size, space_center, nbins):
    """[X*Y*Z, 3] world coordinates, x slowest / z fastest (project_whole.py:28-47)."""
    gx = axis_coords(space_size[0], space_center[0], nbins[0])
    gy = axis_coords(space_size[1], space_center[1], nbins[1])
    gz = axis_coords(space_size[2], space_center[2], nbins[2])
    mx, my, mz = torch.meshgrid(gx, gy, gz, indexing="ij")
    return torch.stack([mx.reshape(-1), my.reshape(-1), mz.reshape(-1)], dim=1)


def camera_tensors(cam):
    """dict of lists / numpy -> fp32 tensors (lib/utils/cameras.py:11-18)."""
    R = torch.as_tensor(np.asarray(cam["R"], dtype=np.float64), dtype=torch.float32).reshape(3, 3)
    T = torch.as_tensor(np.asarray(cam["T"], dtype=np.float64), dtype=torch.float32).reshape(3, 1)
    f = torch.tensor([float(cam["fx"]), float(cam["fy"])], dtype=torch.float32)
    c = torch.tensor([float(cam["cx"]), float(cam["cy"])], dtype=torch.float32)
    k = torch.as_tensor(np.asarray(cam["k"], dtype=np.float64), dtype=torch.float32).reshape(3)
    p = torch.as_tensor(np.asarray(cam["p"], dtype=np.float64), dtype=torch.float32).reshape(2)
    return R, T, f, c, k, p


def _fma32(a, b, c):
    """Exact fp32 fused multiply-add on any host: the product of two floats is exact in float64;
    the sum is formed in float64 with ROUND-TO-ODD (TwoSum error term -> if inexact and the double's
    last mantissa bit is even, step one ulp towards the error), after which the final rounding to
    float32 is the correctly rounded fma (no double-rounding cases).  Independent of CPU / BLAS."""
    p = a.double() * b.double()
    c = c.double()
    s = p + c
    t = s - p
    err = (p - (s - t)) + (c - t)
    odd = (s.view(torch.int64) & 1) == 1
    fix = (err != 0) & ~odd
    toward = torch.where(err > 0, torch.full_like(s, float("inf")), torch.full_like(s, float("-inf")))
    s = torch.where(fix, torch.nextafter(s, toward), s)
    return s.float()


def _mm3_fma(m, x):
    """``torch.mm(m, x)`` for a [R,3] matrix as the reference's CPU sgemm evaluates it in the build
    container: per output a k-ordered chain  fma(m2, x2, fma(m1, x1, m0 * x0))  (found by bit-matching the
    reference's cached grids).  Spelled out because other CPUs / BLAS builds round the same product
    differently (observed: the GPU box's EPYC host differs from the golden vectors by a few ulp), and
    the oracle has to reproduce the GOLDEN arithmetic wherever it runs."""
    rows = []
    for i in range(m.shape[0]):
        acc = m[i, 0] * x[0]
        acc = _fma32(m[i, 1], x[1], acc)
        acc = _fma32(m[i, 2], x[2], acc)
        rows.append(acc)
    return torch.stack(rows, dim=0)


def project_points(pts, cam):
    """World [N,3] -> distorted pixel coordinates [N,2] (lib/utils/cameras.py:30-56).
    No behind-camera test; depth gets +1e-5 (cameras.py:44)."""
    R, T, f, c, k, p = camera_tensors(cam)
    xc = _mm3_fma(R, pts.t() - T)                                   # [3,N]  (torch.mm in the reference)
    y0 = xc[0] / (xc[2] + 1e-5)
    y1 = xc[1] / (xc[2] + 1e-5)
    r = y0 * y0 + y1 * y1
    d = 1 + k[0] * r + k[1] * r * r + k[2] * r * r * r
    u = y0 * d + 2 * p[0] * y0 * y1 + p[1] * (r + 2 * y0 * y0)
    v = y1 * d + 2 * p[1] * y0 * y1 + p[0] * (r + 2 * y1 * y1)
    return torch.stack([f[0] * u + c[0], f[1] * v + c[1]], dim=1)


def sample_grid(pts, cam, cfg, resize_transform):
    """World points -> normalised grid_sample coordinates [N,2] (x = width first).
    lib/models/project_whole.py:49-60: clamp pixels to [-1, max(ori_w, ori_h)], 2x3
    affine (lib/utils/transforms.py:59-63), * [w,h] / IMAGE_SIZE, / [w-1,h-1] * 2 - 1,
    clamp to +-1.1."""
    w, h = cfg.DATASET.HEATMAP_SIZE
    ori = cfg.DATASET.ORI_IMAGE_SIZE
    xy = project_points(pts, cam)
    xy = torch.clamp(xy, -1.0, float(max(ori[0], ori[1])))
    t = torch.as_tensor(np.asarray(resize_transform), dtype=torch.float32)
    homo = torch.cat([xy, torch.ones(xy.shape[0], 1)], dim=1)
    xy = _mm3_fma(t, homo.t())[:2].t()                               # torch.mm in the reference (transforms.py:62)
    xy = xy * torch.tensor([w, h], dtype=torch.float32) / torch.tensor(
        cfg.DATASET.IMAGE_SIZE, dtype=torch.float32)
    g = xy / torch.tensor([w - 1, h - 1], dtype=torch.float32) * 2.0 - 1.0
    return torch.clamp(g, -1.1, 1.1)


def build_sample_grids(pts, cams, cfg, resize_transform):
    """[V,N,2] for all views of one sequence (project_whole.py:75-80)."""
    return torch.stack([sample_grid(pts, cams[c], cfg, resize_transform)
                        for c in range(len(cams))], dim=0)


# --------------------------------------------------------------------------------------
# bilinear back-projection (per frame)
# --------------------------------------------------------------------------------------
def bilinear_mean(heat, grid):
    """heat [V,J,H,W], grid [V,N,2] -> mean over views of bilinear samples, [J,N].
    Restates ``F.grid_sample(bilinear, padding zeros, align_corners=True)`` followed by
    ``torch.mean(dim=0)`` (project_whole.py:83): pixel = (g+1)/2*(size-1); the four taps
    are weighted by the opposite-corner areas; out-of-image taps contribute zero."""
    V, J, H, W = heat.shape
    N = grid.shape[1]
    out = torch.zeros(J, N, dtype=torch.float32)
    for v in range(V):
        ix = (grid[v, :, 0] + 1) * ((W - 1) / 2)
        iy = (grid[v, :, 1] + 1) * ((H - 1) / 2)
        x0 = torch.floor(ix)
        y0 = torch.floor(iy)
        x1 = x0 + 1
        y1 = y0 + 1
        w_nw = (x1 - ix) * (y1 - iy)
        w_ne = (ix - x0) * (y1 - iy)
        w_sw = (x1 - ix) * (iy - y0)
        w_se = (ix - x0) * (iy - y0)
        flat = heat[v].reshape(J, H * W)
        acc = torch.zeros(J, N, dtype=torch.float32)
        for xx, yy, ww in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
            inside = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long()
            val = flat[:, idx] * inside.to(torch.float32)
            acc = acc + val * ww
        out = out + acc
    return out / V


def project_whole(heatmaps, grids_per_frame, voxels):
    """HDN feature cubes [B,J,X,Y,Z] (project_whole.py:62-88): per-frame bilinear mean
    over views, clamp to [0,1]."""
    B, V, J = heatmaps.shape[:3]
    X, Y, Z = voxels
    cubes = torch.stack([bilinear_mean(heatmaps[b], grids_per_frame[b]) for b in range(B)])
    return cubes.clamp(0.0, 1.0).view(B, J, X, Y, Z)


# --------------------------------------------------------------------------------------
# conv stacks on a flat state_dict
# --------------------------------------------------------------------------------------
def _convnd(x, w, b, dim, pad):
    return (F.conv2d if dim == 2 else F.conv1d)(x, w, b, stride=1, padding=pad)


def _bn(sd, key, x):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)


def basic_block(sd, pre, x, dim):
    """Conv(k, pad (k-1)/2) + BN + ReLU (cnns_2d.py:12-23, cnns_1d.py:10-21)."""
    w = sd[pre + ".block.0.weight"]
    x = _convnd(x, w, sd[pre + ".block.0.bias"], dim, (w.shape[-1] - 1) // 2)
    return F.relu(_bn(sd, pre + ".block.1", x))


def res_block(sd, pre, x, dim):
    """relu(BN(conv3(relu(BN(conv3 x)))) + skip(x)); skip = identity or 1x1 conv + BN
    (cnns_2d.py:25-47, cnns_1d.py:24-46)."""
    r = _convnd(x, sd[pre + ".res_branch.0.weight"], sd[pre + ".res_branch.0.bias"], dim, 1)
    r = F.relu(_bn(sd, pre + ".res_branch.1", r))
    r = _convnd(r, sd[pre + ".res_branch.3.weight"], sd[pre + ".res_branch.3.bias"], dim, 1)
    r = _bn(sd, pre + ".res_branch.4", r)
    if (pre + ".skip_con.0.weight") in sd:
        s = _convnd(x, sd[pre + ".skip_con.0.weight"], sd[pre + ".skip_con.0.bias"], dim, 0)
        s = _bn(sd, pre + ".skip_con.1", s)
    else:
        s = x
    return F.relu(r + s)


def upsample_block(sd, pre, x, dim):
    """ConvTranspose(k2, s2) + BN + ReLU (cnns_2d.py:59-71, cnns_1d.py:58-70)."""
    ct = F.conv_transpose2d if dim == 2 else F.conv_transpose1d
    x = ct(x, sd[pre + ".block.0.weight"], sd[pre + ".block.0.bias"], stride=2)
    return F.relu(_bn(sd, pre + ".block.1", x))


def encoder_decoder(sd, pre, x, dim):
    """cnns_2d.py:74-112 / cnns_1d.py:72-109.  The skip adds have no ReLU after them."""
    pool = (lambda t: F.max_pool2d(t, 2, 2)) if dim == 2 else (lambda t: F.max_pool1d(t, 2, 2))
    skip1 = res_block(sd, pre + ".skip_res1", x, dim)
    x = res_block(sd, pre + ".encoder_res1", pool(x), dim)
    skip2 = res_block(sd, pre + ".skip_res2", x, dim)
    x = res_block(sd, pre + ".encoder_res2", pool(x), dim)
    x = res_block(sd, pre + ".mid_res", x, dim)
    x = res_block(sd, pre + ".decoder_res2", x, dim)
    x = upsample_block(sd, pre + ".decoder_upsample2", x, dim) + skip2
    x = res_block(sd, pre + ".decoder_res1", x, dim)
    x = upsample_block(sd, pre + ".decoder_upsample1", x, dim) + skip1
    return x


def trunk(sd, pre, x, dim):
    x = basic_block(sd, pre + ".front_layers.0", x, dim)
    x = res_block(sd, pre + ".front_layers.1", x, dim)
    return encoder_decoder(sd, pre + ".encoder_decoder", x, dim)


def center_net(sd, pre, cubes):
    """z-max then CNN with two heads (cnns_2d.py:173-178)."""
    x = cubes.max(dim=4)[0]
    x = trunk(sd, pre, x, 2)
    heads = []
    for name in ("output_hm", "output_size"):
        h = F.relu(F.conv2d(x, sd[f"{pre}.{name}.0.weight"], sd[f"{pre}.{name}.0.bias"], padding=1))
        heads.append(F.conv2d(h, sd[f"{pre}.{name}.2.weight"], sd[f"{pre}.{name}.2.bias"]))
    return heads[0], heads[1]


def c2c_net(sd, pre, x):
    """1D twin along z (cnns_1d.py:128-132)."""
    x = trunk(sd, pre, x, 1)
    return F.conv1d(x, sd[pre + ".output_hm.weight"], sd[pre + ".output_hm.bias"])


def p2p_net(sd, pre, x):
    """cnns_2d.py:131-135."""
    x = trunk(sd, pre, x, 2)
    return F.conv2d(x, sd[pre + ".output_layer.weight"], sd[pre + ".output_layer.bias"])


def weight_net(sd, pre, joint_features, C):
    """[3,P,J,C,C] -> [3P,J,1]: conv 1->32 k3, BN, maxpool2, ReLU, global avg, MLP, sigmoid
    (lib/models/weight_net.py:48-80; note BN -> MaxPool -> ReLU order :55-60)."""
    x = joint_features.flatten(0, 1)
    n, J = x.shape[0], x.shape[1]
    x = x.reshape(n * J, 1, C, C)
    x = F.conv2d(x, sd[pre + ".heatmap_feature_net.0.weight"], sd[pre + ".heatmap_feature_net.0.bias"], padding=1)
    x = F.relu(F.max_pool2d(_bn(sd, pre + ".heatmap_feature_net.1", x), 2))
    x = x.mean(dim=(2, 3))
    x = F.relu(F.linear(x, sd[pre + ".output.0.weight"], sd[pre + ".output.0.bias"]))
    x = torch.sigmoid(F.linear(x, sd[pre + ".output.2.weight"], sd[pre + ".output.2.bias"]))
    return x.view(n, J, 1)


# --------------------------------------------------------------------------------------
# proposal path (integer / index work: exact)
# --------------------------------------------------------------------------------------
def nms2d(prob_map, max_num):
    """3x3 stride-1 max-pool NMS then top-k (lib/core/proposal.py:13-33).
    Non-maxima become exactly 0.0; maxima keep their value.  Tie rule (torch leaves it
    undefined): value descending, then lowest flat index.  The x index divides by
    ``shape[1]`` = X of the [1,X,Y] map (proposal.py:16-17), as the reference does."""
    B, _, X, Y = prob_map.shape
    pooled = F.max_pool2d(prob_map, 3, 1, 1)
    kept = (prob_map == pooled).float() * prob_map
    flat = kept.reshape(B, -1)
    order = torch.argsort(-flat.double(), dim=1, stable=True)[:, :max_num]
    vals = torch.gather(flat, 1, order)
    ix = torch.div(order, X, rounding_mode="trunc")
    iy = order % X
    return vals, torch.stack([ix, iy], dim=2), order


def hdn_tail(sd, cfg, cubes, hm2d, bbox):
    """NMS/top-k, gathers, C2CNet, z arg-max, proposal packing
    (lib/models/human_detection_net.py:85-104 and ProposalLayer.forward :44-65, eval branch)."""
    B, J = cubes.shape[:2]
    N = cfg.CAPTURE_SPEC.MAX_PEOPLE
    Z = cubes.shape[4]
    conf2d, idx2d, flat = nms2d(hm2d, N)
    bbox_flat = bbox.flatten(2, 3).permute(0, 2, 1)                       # [B, X*Y, 2]
    match_bbox = torch.gather(bbox_flat, 1, flat.unsqueeze(2).repeat(1, 1, 2))
    cols = cubes.flatten(2, 3).permute(0, 2, 1, 3)                        # [B, X*Y, J, Z]
    feat1d = torch.gather(cols, 1, flat.view(B, -1, 1, 1).repeat(1, 1, J, Z))
    hm1d = c2c_net(sd, "pose_net.c2c_net", feat1d.flatten(0, 1)).view(B, N, -1)
    conf1d, idx1d = hm1d.max(dim=2)               # first maximum = lowest z on ties
    idx = torch.cat([idx2d, idx1d.unsqueeze(2)], dim=2)
    conf = conf2d * conf1d
    scale = torch.tensor(cfg.CAPTURE_SPEC.SPACE_SIZE) / (torch.tensor(cfg.CAPTURE_SPEC.VOXELS_PER_AXIS) - 1)
    bias = torch.tensor(cfg.CAPTURE_SPEC.SPACE_CENTER) - torch.tensor(cfg.CAPTURE_SPEC.SPACE_SIZE) / 2.0
    centers = torch.zeros(B, N, 7)
    centers[:, :, 0:3] = idx.float() * scale + bias
    centers[:, :, 3] = (conf > cfg.CAPTURE_SPEC.MIN_SCORE).float() - 1.0
    centers[:, :, 4] = conf
    centers[:, :, 5:7] = match_bbox
    return hm1d, centers, bbox_flat, idx


# --------------------------------------------------------------------------------------
# JLN
# --------------------------------------------------------------------------------------
class IndividualSpec:
    """Constants of lib/models/project_individual.py:14-42."""

    def __init__(self, cfg):
        self.whole_center = torch.tensor(cfg.CAPTURE_SPEC.SPACE_CENTER)
        self.whole_size = torch.tensor(cfg.CAPTURE_SPEC.SPACE_SIZE)
        self.ind_size = torch.tensor(cfg.INDIVIDUAL_SPEC.SPACE_SIZE)
        self.cube = torch.tensor(cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS, dtype=torch.int32)
        self.fine = (self.whole_size / self.ind_size * (self.cube - 1)).int() + 1
        self.scale = (self.fine.float() - 1) / self.whole_size
        self.bias = -self.ind_size / 2.0 / self.whole_size * (self.fine - 1) \
            - self.scale * (self.whole_center - self.whole_size / 2.0)
        g = compute_grid(self.ind_size.tolist(), self.whole_center.tolist(), self.cube.tolist())
        g = g.view(int(self.cube[0]), int(self.cube[1]), int(self.cube[2]), 3)
        self.center_grid = torch.stack([g[:, :, 0, :2].reshape(-1, 2), g[:, 0, :, ::2].reshape(-1, 2),
                                        g[0, :, :, 1:].reshape(-1, 2)])
        self.fine_axes = [axis_coords(float(self.whole_size[a]), float(self.whole_center[a]), int(self.fine[a]))
                          for a in range(3)]

    def fine_points(self, lo, hi):
        """World coordinates of the fine-grid window [lo,hi) (x-major)."""
        ax = [self.fine_axes[a][int(lo[a]):int(hi[a])] for a in range(3)]
        mx, my, mz = torch.meshgrid(*ax, indexing="ij")
        return torch.stack([mx.reshape(-1), my.reshape(-1), mz.reshape(-1)], dim=1)


def person_boxes(spec, centers):
    """Integer window arithmetic of project_individual.py:110-121 for [P,7] proposals.
    round = half-to-even, .int() = truncation; the z margin is always 0."""
    tl = torch.round(centers[:, :3].float() * spec.scale + spec.bias).int()
    offset = tl.float() / (spec.fine - 1) * spec.whole_size - spec.whole_size / 2.0 + spec.ind_size / 2.0
    m = ((1 - centers[:, 5:7]) / 2 * (spec.cube[0:2] - 1)).int()
    m[m < 0] = 0
    m = torch.cat([m, torch.zeros((centers.shape[0], 1), dtype=torch.int32)], dim=1)
    start = torch.where(tl + m >= 0, tl + m, torch.zeros_like(tl))
    end = torch.where(tl + spec.cube - m <= spec.fine, tl + spec.cube - m, spec.fine)
    return tl, offset, start, end


def fine_sample_grid(spec, cfg, cams, resize_transform):
    """The per-sequence cache of project_individual.py:82-94: sampling coordinates of the whole fine
    grid, [V, fx, fy, fz, 2] (164 MB for the Panoptic shape set)."""
    fine = [int(v) for v in spec.fine]
    pts = spec.fine_points(torch.zeros(3, dtype=torch.int64), torch.tensor(fine))
    return build_sample_grids(pts, cams, cfg, resize_transform).view(len(cams), fine[0], fine[1], fine[2], 2)


def project_individual(spec, cfg, heat, centers, cams, resize_transform, fine_grid=None):
    """Per-person cubes [P,J,C,C,C] + offset [P,3] (project_individual.py:96-136).  With
    ``fine_grid`` (``fine_sample_grid``) the window is sliced out of the cached full-space grid as the
    reference does (:127-128); without it the coordinates of the window are recomputed - the
    arithmetic per point is identical."""
    P = centers.shape[0]
    V, J = heat.shape[:2]
    C = [int(c) for c in spec.cube]
    tl, offset, start, end = person_boxes(spec, centers)
    cubes = torch.zeros(P, J, C[0], C[1], C[2])
    for i in range(P):
        if bool((start[i] >= end[i]).any()):
            continue
        if fine_grid is not None:
            grid = fine_grid[:, start[i, 0]:end[i, 0], start[i, 1]:end[i, 1], start[i, 2]:end[i, 2]].reshape(V, -1, 2)
        else:
            grid = build_sample_grids(spec.fine_points(start[i], end[i]), cams, cfg, resize_transform)
        d = (end[i] - start[i]).tolist()
        vals = bilinear_mean(heat, grid).view(J, d[0], d[1], d[2])
        s = (start[i] - tl[i]).tolist()
        cubes[i, :, s[0]:s[0] + d[0], s[1]:s[1] + d[1], s[2]:s[2] + d[2]] = vals
    return cubes.clamp(0.0, 1.0), offset, (tl, start, end)


def triplane_max(cubes):
    """[P,J,C,C,C] -> [3P,J,C,C] = cat(max_z, max_y, max_x) (joint_localization_net.py:80-81)."""
    return torch.cat([cubes.max(dim=4)[0], cubes.max(dim=3)[0], cubes.max(dim=2)[0]])


def soft_argmax(features, center_grid, beta, accumulate=torch.float32):
    """features [3,P,J,C,C] -> poses [3,P,J,2], conf [P] (joint_localization_net.py:20-33).
    ``accumulate=torch.float64`` gives the exact-arithmetic variant used to report the
    reference's own fp32 noise floor."""
    three, P, J = features.shape[:3]
    x = features.reshape(3, P, J, -1, 1).to(accumulate)
    x = F.softmax(beta * x, dim=3)
    conf = x.max(dim=3)[0].squeeze(3).mean(dim=(0, 2))
    g = center_grid.reshape(3, 1, 1, -1, 2).to(accumulate)
    return (x * g).sum(dim=3), conf


def fuse_poses(pose, weights):
    """pose [3,P,J,2], weights [3P,J,1] -> [P,J,3] (joint_localization_net.py:44-62)."""
    w_xy, w_xz, w_yz = torch.chunk(weights, 3)
    xy, xz, yz = pose[0], pose[1], pose[2]
    wx = torch.cat([w_xy, w_xz], dim=2)
    wy = torch.cat([w_xy, w_yz], dim=2)
    wz = torch.cat([w_xz, w_yz], dim=2)
    wx = wx / wx.sum(dim=2, keepdim=True)
    wy = wy / wy.sum(dim=2, keepdim=True)
    wz = wz / wz.sum(dim=2, keepdim=True)
    x = wx[:, :, :1] * xy[:, :, :1] + wx[:, :, 1:] * xz[:, :, :1]
    y = wy[:, :, :1] * xy[:, :, 1:] + wy[:, :, 1:] * yz[:, :, :1]
    z = wz[:, :, :1] * xz[:, :, 1:] + wz[:, :, 1:] * yz[:, :, 1:]
    return torch.cat([x, y, z], dim=2)


# --------------------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------------------
class Oracle:
    """heatmaps [B,V,J,H,W] -> (fused_poses [B,N,J,5], plane_poses [3,B,N,J,2],
    proposal_centers [B,N,7]) exactly as FasterVoxelPoseNet.forward in eval mode
    (lib/models/faster_voxelpose.py:34-105), plus every intermediate in ``self.trace``."""

    def __init__(self, cfg, state_dict):
        self.cfg = cfg
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v
                   for k, v in state_dict.items()}
        self.spec = IndividualSpec(cfg)
        self.whole_pts = compute_grid(cfg.CAPTURE_SPEC.SPACE_SIZE, cfg.CAPTURE_SPEC.SPACE_CENTER,
                                      cfg.CAPTURE_SPEC.VOXELS_PER_AXIS)
        self._grids = {}
        self._fine_grids = {}
        self._sd_cast = {}
        self.trace = {}

    def _cams(self, cameras, seq):
        c = cameras[seq]
        return [c[i] for i in range(len(c))]          # list or int-keyed dict (shelf.py:143-152)

    def hdn(self, heatmaps, meta, cameras, resize_transform):
        cfg = self.cfg
        grids = []
        for seq in meta["seq"]:
            if seq not in self._grids:
                self._grids[seq] = build_sample_grids(self.whole_pts, self._cams(cameras, seq), cfg,
                                                      resize_transform)
            grids.append(self._grids[seq])
        cubes = project_whole(heatmaps, grids, cfg.CAPTURE_SPEC.VOXELS_PER_AXIS)
        hm2d, bbox = center_net(self.sd, "pose_net.center_net", cubes)
        hm1d, centers, bbox_flat, idx = hdn_tail(self.sd, cfg, cubes, hm2d, bbox)
        self.trace.update(cubes=cubes, hm2d=hm2d, bbox=bbox, hm1d=hm1d, topk_index=idx)
        return hm2d, hm1d, centers, bbox_flat

    def _sd(self, dtype):
        if dtype == torch.float32:
            return self.sd
        if dtype not in self._sd_cast:
            self._sd_cast[dtype] = {k: v.to(dtype) if v.is_floating_point() else v for k, v in self.sd.items()}
        return self._sd_cast[dtype]

    def jln(self, meta, heatmaps, centers, mask, cameras, resize_transform, net_dtype=torch.float32):
        """``net_dtype=torch.float64`` evaluates P2PNet / soft-argmax / WeightNet / fusion in
        double on the same fp32 tri-planes: the yardstick for the reference's own fp32
        rounding noise (hard part 1)."""
        cfg = self.cfg
        sd = self._sd(net_dtype)
        B, N = centers.shape[:2]
        J = heatmaps.shape[2]
        C = int(self.spec.cube[0])
        fused = torch.zeros(B, N, J, 3)
        planes = torch.zeros(3, B, N, J, 2)
        per_frame = []
        for i in range(B):
            if int(mask[i].sum()) == 0:
                per_frame.append(None)
                continue
            seq = meta["seq"][i]
            cams = self._cams(cameras, seq)
            if seq not in self._fine_grids:              # cached per sequence, like the reference (:104-106)
                npts = int(self.spec.fine[0]) * int(self.spec.fine[1]) * int(self.spec.fine[2])
                # (jln128's 509 x 509 x 128 fine grid takes minutes through the emulated fma: windows are
                #  recomputed there instead - identical arithmetic per point)
                self._fine_grids[seq] = fine_sample_grid(self.spec, cfg, cams, resize_transform) if npts <= 12_000_000 else None
            cubes, offset, boxes = project_individual(self.spec, cfg, heatmaps[i], centers[i, mask[i]],
                                                      cams, resize_transform, self._fine_grids[seq])
            tri = triplane_max(cubes)
            feat = torch.stack(torch.chunk(p2p_net(sd, "joint_net.conv_net", tri.to(net_dtype)), 3), dim=0)
            pose, conf = soft_argmax(feat, self.spec.center_grid, cfg.NETWORK.BETA, net_dtype)
            off = offset.reshape(-1, 1, 3).to(net_dtype)
            pose[0] += off[:, :, :2]
            pose[1] += off[:, :, ::2]
            pose[2] += off[:, :, 1:]
            w = weight_net(sd, "joint_net.weight_net", feat, C)
            fz = fuse_poses(pose, w)
            fused[i, mask[i]] = fz.to(torch.float32)
            planes[:, i, mask[i]] = pose.to(torch.float32)
            centers[i, mask[i], 4] = conf.to(torch.float32)     # in-place, as the reference (:98)
            per_frame.append(dict(tri=tri, feat=feat, offset=offset, boxes=boxes, weights=w, conf=conf))
        self.trace["jln"] = per_frame
        return fused, planes

    def forward(self, heatmaps, meta, cameras, resize_transform, net_dtype=torch.float32):
        with torch.no_grad():
            heatmaps = heatmaps.to(torch.float32)
            hm2d, hm1d, centers, bbox_flat = self.hdn(heatmaps, meta, cameras, resize_transform)
            mask = centers[:, :, 3] >= 0
            fused, planes = self.jln(meta, heatmaps, centers, mask, cameras, resize_transform, net_dtype)
            B, N = centers.shape[:2]
            J = heatmaps.shape[2]
            fused = torch.cat([fused, centers[:, :, 3:5].reshape(B, N, 1, 2).repeat(1, 1, J, 1)], dim=3)
        return fused, planes, centers


# --------------------------------------------------------------------------------------
# checkpoint layout (key names / shapes), restated from the reference's module definitions
# --------------------------------------------------------------------------------------
def reference_state_dict_shapes(cfg):
    """Ordered {key: zeros tensor} with the key names and shapes of
    ``FasterVoxelPoseNet(cfg).state_dict()`` (485 entries for the shipped configs):
    pose_net.center_net.* (cnns_2d.py:147-171), pose_net.c2c_net.* (cnns_1d.py:112-126),
    joint_net.conv_net.* (cnns_2d.py:115-129), joint_net.weight_net.* (weight_net.py:48-67).
    Module registration order is the reference's ``__init__`` order."""
    J = cfg.DATASET.NUM_JOINTS
    F, Hd = cfg.NETWORK.NUM_CHANNEL_JOINT_FEAT, cfg.NETWORK.NUM_CHANNEL_JOINT_HIDDEN
    out = {}

    def conv(key, cin, cout, k, dim, transposed=False):
        out[key + ".weight"] = torch.zeros(((cin, cout) if transposed else (cout, cin)) + (k,) * dim)
        out[key + ".bias"] = torch.zeros(cout)

    def bn(key, c):
        out[key + ".weight"] = torch.zeros(c)
        out[key + ".bias"] = torch.zeros(c)
        out[key + ".running_mean"] = torch.zeros(c)
        out[key + ".running_var"] = torch.ones(c)
        out[key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def res(pre, cin, cout, dim):
        conv(pre + ".res_branch.0", cin, cout, 3, dim)
        bn(pre + ".res_branch.1", cout)
        conv(pre + ".res_branch.3", cout, cout, 3, dim)
        bn(pre + ".res_branch.4", cout)
        if cin != cout:
            conv(pre + ".skip_con.0", cin, cout, 1, dim)
            bn(pre + ".skip_con.1", cout)

    def up(pre, cin, cout, dim):
        conv(pre + ".block.0", cin, cout, 2, dim, transposed=True)
        bn(pre + ".block.1", cout)

    def trunk(pre, dim):
        conv(pre + ".front_layers.0.block.0", J, 16, 7, dim)
        bn(pre + ".front_layers.0.block.1", 16)
        res(pre + ".front_layers.1", 16, 32, dim)
        ed = pre + ".encoder_decoder"
        res(ed + ".encoder_res1", 32, 64, dim)
        res(ed + ".encoder_res2", 64, 128, dim)
        res(ed + ".mid_res", 128, 128, dim)
        res(ed + ".decoder_res2", 128, 128, dim)
        up(ed + ".decoder_upsample2", 128, 64, dim)
        res(ed + ".decoder_res1", 64, 64, dim)
        up(ed + ".decoder_upsample1", 64, 32, dim)
        res(ed + ".skip_res1", 32, 32, dim)
        res(ed + ".skip_res2", 64, 64, dim)

    trunk("pose_net.center_net", 2)
    for name, c in (("output_hm", 1), ("output_size", 2)):
        conv(f"pose_net.center_net.{name}.0", 32, 32, 3, 2)
        conv(f"pose_net.center_net.{name}.2", 32, c, 1, 2)
    trunk("pose_net.c2c_net", 1)
    conv("pose_net.c2c_net.output_hm", 32, 1, 1, 1)
    trunk("joint_net.conv_net", 2)
    conv("joint_net.conv_net.output_layer", 32, J, 1, 2)
    wn = "joint_net.weight_net"
    conv(wn + ".heatmap_feature_net.0", 1, F, 3, 2)
    bn(wn + ".heatmap_feature_net.1", F)
    out[wn + ".output.0.weight"] = torch.zeros(Hd, F)
    out[wn + ".output.0.bias"] = torch.zeros(Hd)
    out[wn + ".output.2.weight"] = torch.zeros(1, Hd)
    out[wn + ".output.2.bias"] = torch.zeros(1)
    return out


# --------------------------------------------------------------------------------------
# "next" row f-2: input heatmaps rasterised from 2-D detections (precomputed-heatmap path)
# --------------------------------------------------------------------------------------
def compute_human_scale(pose, joints_vis):
    """lib/dataset/JointsDataset.py:197-203."""
    idx = joints_vis > 0.1
    if np.sum(idx) == 0:
        return 0
    minx, maxx = np.min(pose[idx, 0]), np.max(pose[idx, 0])
    miny, maxy = np.min(pose[idx, 1]), np.max(pose[idx, 1])
    return np.clip(np.maximum(maxy - miny, maxx - minx) ** 2, 1.0 / 4 * 96 ** 2, 4 * 96 ** 2)


def generate_input_heatmap(joints, image_size, heatmap_size, sigma):
    """One view: list of [J,>=2] joints in network-image pixels -> [J,H,W] float32
    (JointsDataset.generate_input_heatmap :271-338, eval path: no joints_vis, no augmentation).
    The expressions are kept literally so that numpy's dtype promotion (float32 ``arange``
    against float64 scalars) is the reference's under the numpy installed here."""
    image_size, heatmap_size = np.array(image_size), np.array(heatmap_size)
    num_joints = joints[0].shape[0]
    target = np.zeros((num_joints, heatmap_size[1], heatmap_size[0]), dtype=np.float32)
    feat_stride = image_size / heatmap_size
    for n in range(len(joints)):
        human_scale = 2 * compute_human_scale(joints[n][:, :2] / feat_stride, np.ones(num_joints))
        if human_scale == 0:
            continue
        cur_sigma = sigma * np.sqrt((human_scale / (96.0 * 96.0)))
        tmp_size = cur_sigma * 3
        for joint_id in range(num_joints):
            mu_x = int(joints[n][joint_id][0] / feat_stride[0])
            mu_y = int(joints[n][joint_id][1] / feat_stride[1])
            ul = [int(mu_x - tmp_size), int(mu_y - tmp_size)]
            br = [int(mu_x + tmp_size + 1), int(mu_y + tmp_size + 1)]
            if ul[0] >= heatmap_size[0] or ul[1] >= heatmap_size[1] or br[0] < 0 or br[1] < 0:
                continue
            size = 2 * tmp_size + 1
            x = np.arange(0, size, 1, np.float32)
            y = x[:, np.newaxis]
            x0 = y0 = size // 2
            g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * cur_sigma ** 2))
            g_x = max(0, -ul[0]), min(br[0], heatmap_size[0]) - ul[0]
            g_y = max(0, -ul[1]), min(br[1], heatmap_size[1]) - ul[1]
            img_x = max(0, ul[0]), min(br[0], heatmap_size[0])
            img_y = max(0, ul[1]), min(br[1], heatmap_size[1])
            target[joint_id][img_y[0]:img_y[1], img_x[0]:img_x[1]] = np.maximum(
                target[joint_id][img_y[0]:img_y[1], img_x[0]:img_x[1]], g[g_y[0]:g_y[1], g_x[0]:g_x[1]])
        target = np.clip(target, 0, 1)
    return target


def input_heatmaps_from_pred2d(all_preds, resize_transform, image_size, heatmap_size, sigma):
    """All views of one frame: ``db_rec['pred_pose2d']`` (list over views of lists of [J,>=2]
    arrays in ORIGINAL image pixels) -> [V,J,H,W] (JointsDataset.__getitem__ :144-154:
    affine_transform of every joint, then generate_input_heatmap per view)."""
    t = np.asarray(resize_transform, np.float64)
    out = []
    for preds in all_preds:
        preds = [np.array(p, dtype=np.float64, copy=True) for p in preds]
        for n in range(len(preds)):
            for i in range(len(preds[n])):
                new_pt = np.array([preds[n][i, 0], preds[n][i, 1], 1.0]).T
                preds[n][i, :2] = np.dot(t, new_pt)[:2]                    # utils/transforms.py:53-56
        out.append(torch.from_numpy(generate_input_heatmap(preds, image_size, heatmap_size, sigma)))
    return torch.stack(out, dim=0)


# --------------------------------------------------------------------------------------
# "next" row f-1: Pose-ResNet backbone (lib/models/resnet.py:98-215), functional over a flat state_dict
# --------------------------------------------------------------------------------------
RESNET_SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
               101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}


def _q(t, bf16):
    """bf16 storage emulation: round to bfloat16 (nearest even), keep computing in fp32."""
    return t.bfloat16().float() if bf16 else t


def _bn_affine(sd, key, eps=1e-5):
    sc = sd[key + ".weight"] / torch.sqrt(sd[key + ".running_var"] + eps)
    return sc, sd[key + ".bias"] - sd[key + ".running_mean"] * sc


def pose_resnet(sd, x, num_layers=50, num_deconv=3, bf16=False):
    """x [N,3,H,W] -> heatmaps [N,J,H/4,W/4].  ``bf16=True`` reproduces the numerics of the HIP
    backbone up to summation order: weights and every stored activation rounded to bfloat16, fp32
    accumulation, BatchNorm as one fp32 scale / shift after the conv (folded; the reference applies
    F.batch_norm, identical in exact arithmetic), fp32 heatmaps."""
    block, layers = RESNET_SPEC[num_layers]

    def conv_bn(x, ckey, bnkey, stride=1, pad=0, relu=True, res=None, transposed=False, bias=None):
        w = _q(sd[ckey + ".weight"], bf16)
        if transposed:
            y = F.conv_transpose2d(x, w, None, stride=2, padding=1)
        else:
            y = F.conv2d(x, w, None, stride=stride, padding=pad)
        if bnkey is not None:
            sc, sh = _bn_affine(sd, bnkey)
            if bias is not None:
                sh = sh + bias * sc
            y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        elif bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y

    x = _q(x, bf16)
    x = _q(conv_bn(x, "conv1", "bn1", stride=2, pad=3), bf16)                      # resnet.py:185-187
    x = F.max_pool2d(x, 3, 2, 1)                                                   # :188
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), layers), start=1):
        for b in range(nblocks):
            pre = f"layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            exp = 4 if block == "bottleneck" else 1
            resid = x
            if b == 0 and (stride != 1 or inplanes != planes * exp):                # :133-139
                resid = _q(conv_bn(x, pre + ".downsample.0", pre + ".downsample.1", stride=stride, relu=False), bf16)
            if block == "bottleneck":                                              # :76-95
                h = _q(conv_bn(x, pre + ".conv1", pre + ".bn1"), bf16)
                h = _q(conv_bn(h, pre + ".conv2", pre + ".bn2", stride=stride, pad=1), bf16)
                x = _q(conv_bn(h, pre + ".conv3", pre + ".bn3", res=resid), bf16)
            else:                                                                  # :37-54
                h = _q(conv_bn(x, pre + ".conv1", pre + ".bn1", stride=stride, pad=1), bf16)
                x = _q(conv_bn(h, pre + ".conv2", pre + ".bn2", pad=1, res=resid), bf16)
            inplanes = planes * exp
    for d in range(num_deconv):                                                    # :163-181
        bias = sd.get(f"deconv_layers.{3 * d}.bias")
        x = _q(conv_bn(x, f"deconv_layers.{3 * d}", f"deconv_layers.{3 * d + 1}", transposed=True, bias=bias), bf16)
    return conv_bn(x, "final_layer", None, relu=False, bias=sd["final_layer.bias"])  # :197
