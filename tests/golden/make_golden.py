"""Generate golden vectors from the reference itself (build container only).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Imports ``/root/reference/lib`` (via ``_refimport``), builds ``FasterVoxelPoseNet`` for each
case below with the seeded synthetic weights / heatmaps of
``fvp_synthetic``, runs its eval forward on CPU and stores *data only*:
the recipe (seeds, config name), the reference's outputs and compact digests of its big
intermediates.  Inputs are not stored; they are regenerated from the recipe.

What each .npz holds (all from the reference unless prefixed ``floor_``):
  fused_poses [B,N,J,5], plane_poses [3,B,N,J,2], proposal_centers [B,N,7],
  hm2d [B,X,Y], hm1d [B,N,Z], bbox_match [B,N,2], topk_index [B,N,3] int64,
  grid_digest  : strided sample of the HDN sampling grid [V, n, 2] + its index stride,
  cubes_sub    : cubes[:, :, ::sx, ::sy, :] and fp64 sum / sum-of-squares of all cubes,
  per frame f: jl{f}_tl/start/end int32 [P,3], jl{f}_offset [P,3],
               jl{f}_cubes_sum / _sq (fp64 per person), jl{f}_tri [3P,J,C,C] (first person
               only for the large cases), jl{f}_feat_sub, jl{f}_weights [3P,J], jl{f}_conf [P]
  floor_fused  : the same JLN evaluated in float64 on the reference's fp32 tri-planes
                 (oracle, net_dtype=float64) -> the reference's own fp32 noise floor
  margins      : conditioning of the discrete decisions (top-k gaps, threshold gap,
                 bbox-margin distance to an integer) so a parity failure can be told from
                 an ill-conditioned fixture.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _refimport as R  # noqa: E402
import fvp_synthetic as S  # noqa: E402
import fvp_oracle as O  # noqa: E402

from cases import CASES, make_inputs, make_weights  # noqa: E402


def run_case(ref, case):
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case)
    with R.quiet():
        model = ref.faster_voxelpose.get(cfg).eval()
    sd = make_weights(case, model.state_dict())
    model.load_state_dict(sd)
    B = heat.shape[0]
    N = cfg.CAPTURE_SPEC.MAX_PEOPLE
    J = cfg.DATASET.NUM_JOINTS
    out = {}
    with torch.no_grad(), R.quiet():
        # stage by stage (same calls FasterVoxelPoseNet.forward makes) to capture intermediates
        hdn = model.pose_net
        cubes = hdn.project_layer(heat, meta, cams, rt)
        hm2d, bbox = hdn.center_net(cubes)
        fused, planes, centers, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        hm2d_b, hm1d, centers_hdn, bbox_flat = hdn(heat, meta, cams, rt)
        conf2d, idx2d, flat = ref.proposal.nms2D(hm2d.detach(), N)
    assert torch.equal(hm2d, hm2d_b)
    mask = centers_hdn[:, :, 3] >= 0
    grid = hdn.project_layer.sample_grid[seq].squeeze(1)                      # [V, nbins, 2]
    assert not torch.isnan(grid).any()
    stride = max(1, grid.shape[1] // 1500)
    out["grid_digest"] = grid[:, ::stride].numpy()
    out["grid_stride"] = np.int64(stride)
    X, Y, Z = cfg.CAPTURE_SPEC.VOXELS_PER_AXIS
    sx, sy = max(1, X // 10), max(1, Y // 10)
    out["cubes_sub"] = cubes[:, :, ::sx, ::sy, :].numpy()
    out["cubes_sub_stride"] = np.array([sx, sy], np.int64)
    out["cubes_sum"] = cubes.double().sum(dim=(1, 2, 3, 4)).numpy()
    out["cubes_sq"] = (cubes.double() ** 2).sum(dim=(1, 2, 3, 4)).numpy()
    out["hm2d"] = hm2d[:, 0].numpy()
    out["bbox_map_sub"] = bbox[:, :, ::sx, ::sy].numpy()
    out["hm1d"] = hm1d.numpy()
    out["topk_flat"] = flat.numpy()
    out["conf2d"] = conf2d.numpy()
    out["proposal_centers_hdn"] = centers_hdn.numpy()
    out["proposal_centers"] = centers.numpy()
    out["fused_poses"] = fused.numpy()
    out["plane_poses"] = planes.numpy()

    # JLN intermediates per frame, through the reference's own modules
    jl = model.joint_net
    big = cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS[0] >= 64
    margins = []
    with torch.no_grad(), R.quiet():
        for f in range(B):
            if int(mask[f].sum()) == 0:
                continue
            pc = centers_hdn[f, mask[f]]
            pcubes, offset = jl.project_layer(heat, f, meta, pc, cams, rt)
            pl = jl.project_layer
            tl = torch.round(pc[:, :3].float() * pl.scale + pl.bias).int()
            raw = (1 - pc[:, 5:7]) / 2 * (pl.voxels_per_axis[0:2] - 1)
            m = raw.int()
            m[m < 0] = 0
            m = torch.cat([m, torch.zeros((pc.shape[0], 1), dtype=torch.int32)], dim=1)
            start = torch.where(tl + m >= 0, tl + m, torch.zeros_like(tl))
            end = torch.where(tl + pl.voxels_per_axis - m <= pl.fine_voxels_per_axis,
                              tl + pl.voxels_per_axis - m, pl.fine_voxels_per_axis)
            margins.append(float((raw - torch.round(raw)).abs().min()))
            tri = torch.cat([pcubes.max(dim=4)[0], pcubes.max(dim=3)[0], pcubes.max(dim=2)[0]])
            feat = jl.conv_net(tri)
            jf = torch.stack(torch.chunk(feat, 3), dim=0)
            w = jl.weight_net(jf)
            _, conf = jl.soft_argmax_layer(jf, pl.center_grid)
            out[f"jl{f}_tl"] = tl.numpy()
            out[f"jl{f}_start"] = start.numpy()
            out[f"jl{f}_end"] = end.numpy()
            out[f"jl{f}_offset"] = offset.numpy()
            out[f"jl{f}_cubes_sum"] = pcubes.double().sum(dim=(1, 2, 3, 4)).numpy()
            out[f"jl{f}_cubes_sq"] = (pcubes.double() ** 2).sum(dim=(1, 2, 3, 4)).numpy()
            P = pc.shape[0]
            keep = [0, P, 2 * P] if big else list(range(3 * P))                # person 0's three planes
            out[f"jl{f}_tri_rows"] = np.array(keep, np.int64)
            cs = 5 if big else 1                                               # channel stride kept
            out[f"jl{f}_tri_cstride"] = np.int64(cs)
            out[f"jl{f}_tri"] = tri[keep][:, ::cs].numpy()
            out[f"jl{f}_tri_sum"] = tri.double().sum(dim=(1, 2, 3)).numpy()
            out[f"jl{f}_feat"] = feat[keep][:, ::cs].numpy()
            out[f"jl{f}_feat_sum"] = feat.double().sum(dim=(1, 2, 3)).numpy()
            out[f"jl{f}_weights"] = w[:, :, 0].numpy()
            out[f"jl{f}_conf"] = conf.numpy()
            if f == 0 and not big:
                out["jl0_cubes"] = pcubes.numpy()

    # noise floor: same JLN in float64 (oracle) on the reference's HDN proposals
    orc = O.Oracle(cfg, sd)
    c64 = centers_hdn.clone()
    f64, p64 = orc.jln(meta, heat, c64, mask, cams, rt, net_dtype=torch.float64)
    out["floor_fused"] = f64.numpy()
    out["floor_planes"] = p64.numpy()
    d = (fused[..., :3] - f64)[mask]
    floor = float(d.norm(dim=-1).max()) if d.numel() else 0.0

    # conditioning of discrete decisions
    srt = torch.sort(conf2d, dim=1, descending=True)[0]
    gap2d = float((srt[:, :-1] - srt[:, 1:]).min())
    h1 = torch.sort(hm1d, dim=2, descending=True)[0]
    gap1d = float((h1[..., 0] - h1[..., 1]).min())
    thr = float((centers_hdn[:, :, 4] - cfg.CAPTURE_SPEC.MIN_SCORE).abs().min())
    # the 11th candidate must be clearly below the 10th
    nms = ref.proposal.max_pool2D(hm2d.detach()).reshape(B, -1)
    top = torch.sort(nms, dim=1, descending=True)[0]
    gap_cut = float((top[:, N - 1] - top[:, N]).min())
    out["margins"] = np.array([gap2d, gap1d, thr, gap_cut, min(margins) if margins else 1.0, floor])
    out["valid"] = mask.numpy()
    print("   conf:", np.round(centers_hdn[:, :, 4].numpy(), 3).tolist())
    print(f"{case:22s} valid/frame {mask.sum(1).tolist()}  top-k gap {gap2d:.2e}  cut gap {gap_cut:.2e}  "
          f"z gap {gap1d:.2e}  thr gap {thr:.2e}  bbox-int gap {out['margins'][4]:.2e}  "
          f"fp32-vs-fp64 floor {floor:.2e} mm")
    assert min(gap2d, gap1d, thr, gap_cut, out["margins"][4]) > 1e-4, "ill-conditioned fixture"
    return out


def main():
    ref = R.import_reference()
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for case in CASES:
        if only and case not in only:
            continue
        out = run_case(ref, case)
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, **out)
        print(f"   -> {os.path.relpath(path, ROOT)}  {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
