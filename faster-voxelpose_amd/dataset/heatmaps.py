"""Input heatmaps from 2-D detections on the GPU -- the reference's "precomputed-heatmap path"
(Shelf / Campus: ``TEST_HEATMAP_SRC = 'pred'``), which it rasterises on the CPU per sample in
``lib/dataset/JointsDataset.py`` (``__getitem__`` :144-154, ``generate_input_heatmap`` :271-338).

Host side (this file): the per-joint ``affine_transform`` into network-image pixels (the
reference's own numpy expression, utils/transforms.py:53-56 -- a handful of float64 dot products)
and packing into one tensor.  Device side: ``fvp_rasterise_heatmaps`` (csrc/fvp_heatmap.hip).
"""
import ctypes as C

import numpy as np
import torch

from .. import _capi as capi


def generate_input_heatmaps(pred_pose2d, resize_transform, cfg, sigma=None, device=None, channels_last=False,
                            _lib=None):
    """``pred_pose2d``: one frame = list over views of lists of ``[J, >=2]`` arrays (ORIGINAL image
    pixels, ``db_rec['pred_pose2d']``), or a list of such frames.  Returns ``[V,J,H,W]`` (or
    ``[B,V,J,H,W]``) float32 on ``device``; with ``channels_last=True`` also the ``[.., H*W, JP]``
    staging copy the projection kernels read."""
    batched = len(pred_pose2d) > 0 and len(pred_pose2d[0]) > 0 and isinstance(pred_pose2d[0][0], (list, tuple))
    frames = pred_pose2d if batched else [pred_pose2d]
    device = torch.device(device if device is not None else cfg.DEVICE)
    lib = _lib if _lib is not None else capi.load()
    if _lib is None and device.type != "cuda":
        raise capi.FvpError("generate_input_heatmaps runs on the GPU only (no CPU fallback)")
    J = cfg.DATASET.NUM_JOINTS
    W, H = cfg.DATASET.HEATMAP_SIZE
    fs = np.array(cfg.DATASET.IMAGE_SIZE) / np.array(cfg.DATASET.HEATMAP_SIZE)       # feat_stride (:275)
    sigma = float(cfg.NETWORK.SIGMA if sigma is None and hasattr(cfg.NETWORK, "SIGMA") else (3 if sigma is None else sigma))
    t = np.asarray(resize_transform.detach().cpu() if isinstance(resize_transform, torch.Tensor) else resize_transform,
                   dtype=np.float64).reshape(2, 3)
    V = len(frames[0])
    P = max([len(v) for f in frames for v in f] + [1])
    joints = np.zeros((len(frames), V, P, J, 2), np.float64)
    counts = np.zeros((len(frames), V), np.int32)
    for b, frame in enumerate(frames):
        assert len(frame) == V, "every frame needs the same number of views"
        for v, preds in enumerate(frame):
            counts[b, v] = len(preds)
            for n, person in enumerate(preds):
                person = np.asarray(person, np.float64)
                assert person.shape[0] == J and person.shape[1] >= 2
                for i in range(J):                                               # JointsDataset.py:149-151
                    joints[b, v, n, i] = np.dot(t, np.array([person[i, 0], person[i, 1], 1.0]).T)[:2]
    nimg = len(frames) * V
    jd = torch.from_numpy(joints).to(device)
    cd = torch.from_numpy(counts).to(device)
    out = torch.empty((len(frames), V, J, H, W), dtype=torch.float32, device=device)
    JP = (J + 3) // 4 * 4
    cl = torch.empty((len(frames), V, H * W, JP), dtype=torch.float32, device=device) if channels_last else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if device.type == "cuda" else None
    rc = lib.fvp_rasterise_heatmaps(C.c_void_p(jd.data_ptr()), C.c_void_p(cd.data_ptr()), nimg, P, J, W, H,
                                    float(fs[0]), float(fs[1]), sigma, C.c_void_p(out.data_ptr()),
                                    C.c_void_p(cl.data_ptr()) if cl is not None else None, JP, stream)
    capi.check(lib, rc, "fvp_rasterise_heatmaps")
    if not batched:
        out = out[0]
        cl = cl[0] if cl is not None else None
    return (out, cl) if channels_last else out
