"""Host-side image-geometry helpers for the inference hot path.

Mirrors the *interface* of the reference's ``lib/utils/transforms.py`` for the three
symbols the hot path needs (``get_scale`` :81, ``get_affine_transform`` :15,
``affine_transform_pts_cuda`` :59).  Nothing here runs per frame: the 2x3
``resize_transform`` is a per-dataset constant that is handed to the HIP kernels as six
floats (see ``include/fvp.h`` ``FvpGeom``).

OpenCV is not a dependency: the three-point affine is solved directly in float64.
"""
import numpy as np
import torch


def get_scale(image_size, resized_size):
    """Letter-box scale in units of 200 px (reference transforms.py:81-93)."""
    w, h = image_size
    w_resized, h_resized = resized_size
    if w / w_resized < h / h_resized:
        w_pad, h_pad = h / h_resized * w_resized, h
    else:
        w_pad, h_pad = w, w / w_resized * h_resized
    return np.array([w_pad / 200.0, h_pad / 200.0], dtype=np.float32)


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def _solve_affine(src, dst):
    """2x3 affine A with A @ [x, y, 1] = dst for three point pairs (float64)."""
    m = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    sol = np.linalg.solve(m, dst.astype(np.float64))          # [3, 2]
    return sol.T.copy()                                        # [2, 3]


def get_affine_transform(center, scale, rot, output_size,
                         shift=np.array([0, 0], dtype=np.float32), inv=0):
    """Centre/scale/rotation -> 2x3 affine (reference transforms.py:15-49)."""
    if isinstance(scale, torch.Tensor):
        scale = np.array(scale.cpu())
    if isinstance(center, torch.Tensor):
        center = np.array(center.cpu())
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale])
    scale_tmp = np.asarray(scale) * 200.0
    src_w, src_h = scale_tmp[0], scale_tmp[1]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    if src_w >= src_h:
        p = [0, src_w * -0.5]
        dst_dir = np.array([0, dst_w * -0.5], np.float32)
    else:
        p = [src_h * -0.5, 0]
        dst_dir = np.array([dst_h * -0.5, 0], np.float32)
    src_dir = [p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs]

    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def get_resize_transform(ori_image_size, image_size):
    """The per-dataset constant the reference builds in JointsDataset.py:51-56."""
    c = np.array([ori_image_size[0] / 2.0, ori_image_size[1] / 2.0])
    s = get_scale((ori_image_size[0], ori_image_size[1]), image_size)
    return get_affine_transform(c, s, 0, image_size)


def affine_transform_pts_cuda(pts, t):
    """[N,2] points through a 2x3 affine on the points' device (reference :59-63)."""
    ones = torch.ones(pts.shape[0], 1, device=pts.device, dtype=pts.dtype)
    return (t @ torch.cat([pts, ones], dim=1).t())[:2].t()
