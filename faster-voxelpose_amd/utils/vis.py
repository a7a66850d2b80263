"""Debug visualisation of the hot path's outputs -- counterpart of the reference's
``lib/utils/vis.py`` (``test_vis_all`` :47, ``save_2d_planes`` :140, ``save_image_with_poses`` :222,
``save_heatmaps`` :281): same entry points, arguments, output folders and file names, drawn with
matplotlib only (the reference needs OpenCV and torchvision for two of the three views; neither is a
dependency here).  Host-side tooling: nothing in this file is on the timed path.

Figures:
  ``<dir>/2d_planes/<name>.png``             per frame: 3-D skeletons + the xy / xz / yz plane estimates with
                                             the proposals' bounding boxes (ground truth in red if ``meta`` has it)
  ``<dir>/image_with_poses/<name>_view_k.jpg`` the fused skeletons projected into every camera image
  ``<dir>/heatmaps/<name>_view_k.jpg``       image | joint heatmaps blended over the image, one row per frame
"""
import os

import matplotlib

matplotlib.use("Agg")
import numpy as np  # noqa: E402
from matplotlib import pyplot as plt  # noqa: E402
from matplotlib.patches import Rectangle  # noqa: E402

# skeleton topologies of the three joint sets the reference handles (data: joint index pairs)
LIMBS17 = [[0, 1], [0, 2], [1, 2], [1, 3], [2, 4], [3, 5], [4, 6], [5, 7], [7, 9], [6, 8], [8, 10], [5, 11], [11, 13],
           [13, 15], [6, 12], [12, 14], [14, 16], [5, 6], [11, 12]]                              # COCO-17
LIMBS14 = [[0, 1], [1, 2], [3, 4], [4, 5], [2, 3], [6, 7], [7, 8], [9, 10], [10, 11], [2, 8], [3, 9], [8, 12], [9, 12],
           [12, 13]]                                                                             # Shelf / Campus
LIMBS15 = [[0, 1], [0, 2], [0, 3], [3, 4], [4, 5], [0, 9], [9, 10], [10, 11], [2, 6], [2, 12], [6, 7], [7, 8], [12, 13],
           [13, 14]]                                                                             # Panoptic
_LIMBS = {17: LIMBS17, 14: LIMBS14, 15: LIMBS15}
colors = ["b", "g", "c", "y", "m", "orange", "pink", "royalblue", "lightgreen", "gold"]
idx_list = {"xy": [0, 1], "xz": [0, 2], "yz": [1, 2]}


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _limbs(num_joints):
    if num_joints not in _LIMBS:
        raise ValueError(f"no skeleton topology for {num_joints} joints (known: 14, 15, 17)")
    return _LIMBS[num_joints]


def is_valid_coord(joint, width, height):
    return 0 <= joint[0] < width and 0 <= joint[1] < height


def _segments(pose, dims):
    """[limb, 2 endpoints, len(dims)] coordinates of one skeleton."""
    pose = _np(pose)
    limbs = np.asarray(_limbs(len(pose)))
    return pose[limbs][:, :, dims]


def _style(person, vis_pair):
    """Predictions: one colour per person; ground truth: red, dashed where an endpoint is not visible."""
    base = dict(lw=1.5, marker="o", markerfacecolor="w", markersize=2, markeredgewidth=1)
    if vis_pair is None:
        return dict(base, c=colors[person % len(colors)])
    return dict(base, c="r", ls="-" if min(vis_pair) > 0.1 else "--")


def vis_3d_poses(ax, num_person, poses, poses_vis=None):
    for n in range(int(num_person)):
        seg = _segments(poses[n], [0, 1, 2])
        limbs = _limbs(len(_np(poses[n])))
        for k, s in enumerate(seg):
            vp = None if poses_vis is None else (float(_np(poses_vis[n])[limbs[k][0]]), float(_np(poses_vis[n])[limbs[k][1]]))
            ax.plot(s[:, 0], s[:, 1], s[:, 2], **_style(n, vp))


def vis_2d_poses(ax, num_person, poses, poses_vis=None, plane_type="xy"):
    # plane estimates are already 2-D (columns 0, 1); ground truth is 3-D and gets projected on the plane
    dims = idx_list[plane_type] if poses_vis is not None else [0, 1]
    for n in range(int(num_person)):
        seg = _segments(poses[n], dims)
        limbs = _limbs(len(_np(poses[n])))
        for k, s in enumerate(seg):
            vp = None if poses_vis is None else (float(_np(poses_vis[n])[limbs[k][0]]), float(_np(poses_vis[n])[limbs[k][1]]))
            ax.plot(s[:, 0], s[:, 1], **_style(n, vp))


def vis_2d_bbox(ax, config, proposal_centers, plane_type="xy"):
    """The cube window of every valid proposal on one plane: bbox_w/h (columns 5, 6) scale the x / y extent of
    the individual space, z always spans it fully."""
    a0, a1 = idx_list[plane_type]
    space = np.asarray(config.INDIVIDUAL_SPEC.SPACE_SIZE, dtype=np.float64)
    pc = _np(proposal_centers)
    for row in pc:
        if row[3] < 0:
            continue
        extent = np.array([row[5] * space[0], row[6] * space[1], space[2]])
        corner = row[:3] - extent / 2
        ax.add_patch(Rectangle((corner[a0], corner[a1]), extent[a0], extent[a1], fill=False, edgecolor="red", linewidth=1))


def _out_path(prefix, folder, suffix):
    d = os.path.join(os.path.dirname(prefix), folder)
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, os.path.basename(prefix) + suffix)


def save_2d_planes(config, meta, fused_poses, plane_poses, proposal_centers, prefix):
    file_name = _out_path(prefix, "2d_planes", ".png")
    fused, planes, pc = _np(fused_poses), _np(plane_poses), _np(proposal_centers)
    B = fused.shape[0]
    mask = pc[:, :, 3] >= 0
    fig = plt.figure(figsize=(16.0, 4.0 * B))
    fig.subplots_adjust(left=0.05, right=0.95, bottom=0.05, top=0.95, wspace=0.2, hspace=0.15)
    titles = ["3d pose", "xy projection", "xz projection", "yz projection"]
    has_gt = "joints_3d" in meta
    for i in range(B):
        gt = (int(meta["num_person"][i]), meta["joints_3d"][i], meta["joints_3d_vis"][i]) if has_gt else None
        for col in range(4):
            ax = fig.add_subplot(B, 4, 4 * i + col + 1, **({"projection": "3d"} if col == 0 else {}))
            if col == 0:
                if gt:
                    vis_3d_poses(ax, gt[0], gt[1], poses_vis=gt[2])
                vis_3d_poses(ax, int(mask[i].sum()), fused[i][mask[i]])
            else:
                plane = ("xy", "xz", "yz")[col - 1]
                if gt:
                    vis_2d_poses(ax, gt[0], gt[1], poses_vis=gt[2], plane_type=plane)
                vis_2d_poses(ax, int(mask[i].sum()), planes[col - 1][i][mask[i]])
                vis_2d_bbox(ax, config, pc[i], plane_type=plane)
            if i == 0:
                ax.set_title(titles[col], fontdict={"weight": "normal", "size": 15})
    fig.savefig(file_name)
    plt.close(fig)
    return file_name


def project_pose_np(points, cam):
    """World points [N,3] (mm) -> distorted pixel coordinates [N,2] in the ORIGINAL image: the camera model of the
    reference's utils/cameras.py (pinhole + 3 radial + 2 tangential terms, depth + 1e-5), in float64 on the host."""
    R = np.asarray(cam["R"], np.float64).reshape(3, 3)
    T = np.asarray(cam["T"], np.float64).reshape(3, 1)
    k = np.asarray(cam["k"], np.float64).reshape(3)
    p = np.asarray(cam["p"], np.float64).reshape(2)
    xc = R @ (np.asarray(points, np.float64).T - T)
    y = xc[:2] / (xc[2] + 1e-5)
    r = (y * y).sum(0)
    radial = 1 + k[0] * r + k[1] * r ** 2 + k[2] * r ** 3
    u = y[0] * radial + 2 * p[0] * y[0] * y[1] + p[1] * (r + 2 * y[0] ** 2)
    v = y[1] * radial + 2 * p[1] * y[0] * y[1] + p[0] * (r + 2 * y[1] ** 2)
    return np.stack([float(cam["fx"]) * u + float(cam["cx"]), float(cam["fy"]) * v + float(cam["cy"])], axis=1)


def _to_rgb01(img_chw, bgr=True):
    """[3,H,W] network input (BGR like the reference's loader, arbitrary range) -> [H,W,3] RGB in [0,1]."""
    a = _np(img_chw).astype(np.float64)
    a = (a - a.min()) / (a.max() - a.min() + 1e-5)
    a = a[::-1] if bgr else a
    return np.transpose(a, (1, 2, 0))


def save_image_with_poses(config, images, poses, meta, cameras, resize_transform, prefix):
    """images [B,V,3,H,W]; poses = fused_poses [B,N,J,5].  One file per view, frames stacked vertically."""
    imgs, poses = _np(images), _np(poses)
    B, V, _, H, W = imgs.shape
    rt = _np(resize_transform).astype(np.float64).reshape(2, 3)
    limbs = _limbs(poses.shape[2])
    files = []
    for c in range(V):
        fig, axes = plt.subplots(B, 1, figsize=(W / 100.0, B * H / 100.0), dpi=100, squeeze=False)
        fig.subplots_adjust(left=0, right=1, bottom=0, top=1, hspace=0)
        for i in range(B):
            ax = axes[i, 0]
            ax.imshow(_to_rgb01(imgs[i, c]), extent=(0, W, H, 0))
            ax.set_xlim(0, W)
            ax.set_ylim(H, 0)
            ax.axis("off")
            cam = cameras[meta["seq"][i]][c]
            for n in range(poses.shape[1]):
                if poses[i, n, 0, 4] < config.CAPTURE_SPEC.MIN_SCORE:
                    continue
                px = project_pose_np(poses[i, n, :, :3], cam)
                px = px @ rt[:, :2].T + rt[:, 2]                       # original image -> network image
                ok = [is_valid_coord(q, W, H) for q in px]
                col = colors[n % len(colors)]
                ax.scatter(px[ok, 0], px[ok, 1], s=40, c=col, zorder=3)
                for a, b in limbs:
                    if ok[a] and ok[b]:
                        ax.plot(px[[a, b], 0], px[[a, b], 1], c=col, lw=3, zorder=2)
        name = _out_path(prefix, "image_with_poses", f"_view_{c + 1}.jpg")
        fig.savefig(name)
        plt.close(fig)
        files.append(name)
    return files


def save_heatmaps(batch_images, batch_heatmaps, prefix):
    """batch_images [B,V,3,H,W], batch_heatmaps [B,V,J,h,w]: per view a grid with one row per frame:
    the image, then every joint's heatmap (jet colour map, 70 %) blended over the image (30 %)."""
    imgs, heat = _np(batch_images), _np(batch_heatmaps)
    B, V, _, H, W = imgs.shape
    J = heat.shape[2]
    jet = matplotlib.colormaps["jet"]
    files = []
    for c in range(V):
        lo, hi = float(imgs[:, c].min()), float(imgs[:, c].max())
        grid = np.zeros((B * H, (J + 1) * W, 3), np.float64)
        for i in range(B):
            rgb = np.transpose(((imgs[i, c].astype(np.float64) - lo) / (hi - lo + 1e-5))[::-1], (1, 2, 0))
            grid[i * H:(i + 1) * H, :W] = rgb
            for j in range(J):
                hm = np.clip(heat[i, c, j], 0.0, 1.0)
                ys = np.minimum((np.arange(H) * hm.shape[0]) // H, hm.shape[0] - 1)       # nearest-neighbour upsampling
                xs = np.minimum((np.arange(W) * hm.shape[1]) // W, hm.shape[1] - 1)
                grid[i * H:(i + 1) * H, (j + 1) * W:(j + 2) * W] = 0.7 * jet(hm[ys][:, xs])[..., :3] + 0.3 * rgb
        name = _out_path(prefix, "heatmaps", f"_view_{c + 1}.jpg")
        plt.imsave(name, np.clip(grid, 0.0, 1.0))
        files.append(name)
    return files


def _vis_all(vis_types, heatmap_src, config, meta, cameras, resize_transform, inputs, input_heatmaps, fused_poses, plane_poses,
             proposal_centers, prefix, phase):
    if "2d_planes" in vis_types:
        save_2d_planes(config, meta, fused_poses, plane_poses, proposal_centers, prefix)
    if "image_with_poses" in vis_types:
        save_image_with_poses(config, inputs, fused_poses, meta, cameras, resize_transform, prefix)
    if "heatmaps" in vis_types:
        if (phase == "train" and heatmap_src != "image") or (phase == "test" and heatmap_src == "pred"):
            raise ValueError("cannot visualize heatmaps" + (" only given 2D predictions" if phase == "test" else ""))
        save_heatmaps(inputs, input_heatmaps, prefix)


def train_vis_all(config, meta, cameras, resize_transform, inputs, input_heatmaps, fused_poses, plane_poses, proposal_centers,
                  prefix):
    _vis_all(config.TRAIN.VIS_TYPE, config.DATASET.TRAIN_HEATMAP_SRC, config, meta, cameras, resize_transform, inputs,
             input_heatmaps, fused_poses, plane_poses, proposal_centers, prefix, "train")


def test_vis_all(config, meta, cameras, resize_transform, inputs, input_heatmaps, fused_poses, plane_poses, proposal_centers,
                 prefix):
    _vis_all(config.TEST.VIS_TYPE, config.DATASET.TEST_HEATMAP_SRC, config, meta, cameras, resize_transform, inputs,
             input_heatmaps, fused_poses, plane_poses, proposal_centers, prefix, "test")


test_vis_all.__test__ = False      # (not a pytest test despite the reference's name)
