"""``nms2D`` -- drop-in for the reference's ``lib/core/proposal.py:27-33``.

3x3 max-pool NMS followed by top-k, in one HIP kernel (``fvp_nms_topk``).  Returns
``(values [B,N], index int64 [B,N,2], flat index int64 [B,N])`` like the reference,
including its quirk of unravelling both coordinates with ``shape[1]`` (:16-17).  Ties, which
``torch.topk`` leaves unspecified, resolve to the lowest flat index.
"""
import ctypes as C

import torch

from .. import _capi as capi


def nms2D(prob_map, max_num, _lib=None):
    lib = _lib if _lib is not None else capi.load()
    if _lib is None and prob_map.device.type != "cuda":
        raise capi.FvpError("nms2D runs on the GPU only (no CPU fallback)")
    B, _, X, Y = prob_map.shape
    pm = prob_map.contiguous().float()
    dev = pm.device
    vals = torch.empty((B, max_num), device=dev)
    idx = torch.empty((B, max_num, 2), dtype=torch.int64, device=dev)
    flat = torch.empty((B, max_num), dtype=torch.int64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
    capi.check(lib, lib.fvp_nms_topk(C.c_void_p(pm.data_ptr()), B, X, Y, max_num, C.c_void_p(vals.data_ptr()),
                                     C.c_void_p(idx.data_ptr()), C.c_void_p(flat.data_ptr()), stream), "fvp_nms_topk")
    return vals, idx, flat
