"""Checkpoint ingestion for the hot-path model (reference: ``lib/utils/utils.py`` --
``save_checkpoint`` :88-98 writes ``model_best.pth.tar`` as a BARE state_dict without the
backbone's entries, ``checkpoint.pth.tar`` as ``{'state_dict': ..., 'epoch': ..., optimizers}``;
``run/validate.py:76-81`` loads the former with ``model.load_state_dict(torch.load(file))``)."""
import numpy as np
import torch


def _numpy_scalar_globals():
    """What a pickled numpy scalar / small array needs besides tensors: ``run/train.py`` stores
    ``'precision': best_precision`` (an ``np.float64`` from ``np.mean``) next to the state_dict, which the
    restricted unpickler of ``weights_only=True`` rejects unless these are allow-listed."""
    names = []
    try:
        from numpy._core import multiarray as ma
    except ImportError:                                   # numpy < 2
        from numpy.core import multiarray as ma
    names += [ma.scalar, ma._reconstruct, np.dtype, np.ndarray, np.float64, np.float32, np.int64, np.int32, np.bool_]
    names += [type(np.dtype(t)) for t in (np.float64, np.float32, np.int64, np.int32, np.bool_)]
    return names


def read_state_dict(path, map_location="cpu"):
    """Either file flavour -> flat ``{key: tensor}`` of the voxel model (backbone entries dropped,
    a leading ``module.`` of DataParallel checkpoints stripped)."""
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        obj = torch.load(path, map_location=map_location, weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and not torch.is_tensor(obj["state_dict"]):
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{path}: neither a state_dict nor a checkpoint with a 'state_dict' entry")
    out = {}
    for k, v in obj.items():
        k = k[7:] if k.startswith("module.") else k
        if "backbone" in k:
            continue
        out[k] = v
    return out


def load_model_file(model, path, strict=True):
    """``model.load_state_dict`` with a readable report of missing / unexpected keys."""
    sd = read_state_dict(path)
    want = model.state_dict()
    missing = [k for k in want if k not in sd]
    extra = [k for k in sd if k not in want]
    bad = [k for k in sd if k in want and tuple(sd[k].shape) != tuple(want[k].shape)]
    if strict and (missing or extra or bad):
        raise ValueError(f"{path} does not match the model: {len(missing)} missing (e.g. {missing[:3]}), "
                         f"{len(extra)} unexpected (e.g. {extra[:3]}), {len(bad)} shape mismatches (e.g. {bad[:3]})")
    model.load_state_dict({k: v for k, v in sd.items() if k in want and k not in bad}, strict=False)
    return dict(missing=missing, unexpected=extra, shape_mismatch=bad)
