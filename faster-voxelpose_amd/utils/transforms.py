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
    """Size of the letter-boxed source window in units of 200 px (interface of the reference's
    transforms.py:81): the original image padded to the aspect ratio of the resized one."""
    ow, oh = float(image_size[0]), float(image_size[1])
    rw, rh = float(resized_size[0]), float(resized_size[1])
    # the axis with the larger original/resized ratio keeps its length, the other is padded
    if ow / rw < oh / rh:
        padded = (oh / rh * rw, oh)
    else:
        padded = (ow, ow / rw * rh)
    return (np.array(padded, dtype=np.float64) / 200.0).astype(np.float32)


def _triangle(origin, arm):
    """float32 triangle (origin, origin + arm, perpendicular completion of that edge at its
    far end) - the three control points the reference hands to OpenCV (transforms.py:36-44);
    the completion is done in float32 as there, which is why the result is a similarity only
    up to float32 rounding of the third vertex."""
    tri = np.empty((3, 2), dtype=np.float32)
    tri[0] = origin
    tri[1] = np.asarray(origin, dtype=np.float64) + arm
    edge = tri[0] - tri[1]
    tri[2] = tri[1] + np.array([-edge[1], edge[0]], dtype=np.float32)
    return tri


def _affine_from_triangles(src, dst):
    """2x3 affine A with A @ [x, y, 1] = dst for the three vertex pairs, solved in float64."""
    m = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(m, dst.astype(np.float64)).T.copy()


def get_affine_transform(center, scale, rot, output_size,
                         shift=np.array([0, 0], dtype=np.float32), inv=0):
    """Centre / scale (units of 200 px) / rotation (degrees) -> 2x3 affine into an
    ``output_size`` window (interface of the reference's transforms.py:15).

    The longer source side is mapped onto the same side of the output: a half-side "arm"
    (pointing up for a wide source, left for a tall one) is attached to the source anchor
    ``center + shift * size`` - rotated by ``rot`` - and to the output centre - unrotated; the two
    triangles spanned by anchor, arm and the arm's perpendicular define the affine."""
    size = np.asarray(scale.cpu() if isinstance(scale, torch.Tensor) else scale)
    if size.ndim == 0:
        size = np.array([size, size])
    size = size * 200.0
    anchor = np.asarray(center.cpu() if isinstance(center, torch.Tensor) else center) + size * shift
    out_w, out_h = output_size[0], output_size[1]
    wide = size[0] >= size[1]
    half_src = float(-0.5 * (size[0] if wide else size[1]))
    half_dst = float(np.float32(-0.5 * (out_w if wide else out_h)))
    axis = 1j if wide else 1.0                              # unit vector of the arm: +y or +x
    turn = complex(np.cos(np.pi * rot / 180), np.sin(np.pi * rot / 180))
    arm_src = half_src * axis * turn
    arm_dst = half_dst * axis
    src = _triangle(anchor, np.array([arm_src.real, arm_src.imag]))
    dst = _triangle([out_w * 0.5, out_h * 0.5], np.array([arm_dst.real, arm_dst.imag]))
    return _affine_from_triangles(dst, src) if inv else _affine_from_triangles(src, dst)


def get_resize_transform(ori_image_size, image_size):
    """The per-dataset constant the reference builds in JointsDataset.py:51-56."""
    c = np.array([ori_image_size[0] / 2.0, ori_image_size[1] / 2.0])
    s = get_scale((ori_image_size[0], ori_image_size[1]), image_size)
    return get_affine_transform(c, s, 0, image_size)


def affine_transform_pts_cuda(pts, t):
    """[N,2] points through a 2x3 affine on the points' device (reference :59-63)."""
    ones = torch.ones(pts.shape[0], 1, device=pts.device, dtype=pts.dtype)
    return (t @ torch.cat([pts, ones], dim=1).t())[:2].t()
