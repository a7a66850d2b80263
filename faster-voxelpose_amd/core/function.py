"""Validation loop over the hot path -- counterpart of the reference's ``validate``
(``lib/core/function.py:117-174``): batches in, ``fused_poses`` out, optionally scored.

A batch is a dict with ``meta`` (at least ``{'seq': [...]}``) and either ``input_heatmaps``
``[B,V,J,H,W]`` (the ``'image'`` / ``'gt'`` sources, heatmaps already computed) or
``pred_pose2d`` (the ``'pred'`` source of Shelf / Campus: per frame, per view, a list of
``[J,>=2]`` detections in original-image pixels), which is rasterised on the GPU
(``dataset.heatmaps.generate_input_heatmaps``).  With ``depth > 1`` batches are kept in flight
on separate HIP streams (``PipelinedForward``)."""
import time

import torch

from ..dataset.heatmaps import generate_input_heatmaps
from ..models.faster_voxelpose import PipelinedForward


def validate(config, model, batches, cameras, resize_transform, evaluate=None, depth=1, log=None):
    """Returns ``(metric, all_fused_poses [sum B, N, J, 5], info)``; ``evaluate`` is a callable
    ``all_fused_poses (cpu numpy list per frame) -> dict`` with a ``'metric'`` entry (e.g.
    ``functools.partial(metrics.evaluate_panoptic, gt_joints=..., gt_vis=...)``)."""
    model.eval()
    dev = torch.device(config.DEVICE)
    rt = torch.as_tensor(resize_transform, dtype=torch.float, device=dev)
    pipe = PipelinedForward(model, depth=depth) if depth > 1 else None
    outs, frames = [], 0
    t0 = time.perf_counter()
    with torch.no_grad():
        for i, batch in enumerate(batches):
            if batch.get("input_heatmaps") is not None:
                heat = batch["input_heatmaps"].to(dev)
            else:
                heat = generate_input_heatmaps(batch["pred_pose2d"], resize_transform, config, device=dev)
            kwargs = dict(meta=batch["meta"], input_heatmaps=heat, cameras=cameras, resize_transform=rt)
            if pipe is not None:
                (fused, _, _, _, _), _ = pipe.submit(**kwargs)
            else:
                fused, _, _, _, _ = model(**kwargs)
            outs.append(fused)
            frames += heat.shape[0]
            if log is not None and (i % getattr(config, "PRINT_FREQ", 100) == 0):
                log(f"Test: [{i}] {frames} frames enqueued")
        if pipe is not None:
            pipe.synchronize()
        torch.cuda.synchronize(dev) if dev.type == "cuda" else None
    dt = time.perf_counter() - t0
    all_fused = torch.cat(outs, dim=0) if outs else torch.empty(0)
    info = dict(frames=frames, seconds=dt, frames_per_second=frames / dt if dt > 0 else float("inf"))
    if evaluate is None:
        return 0.0, all_fused, info
    result = evaluate([p for p in all_fused.detach().cpu().numpy()])
    info["evaluation"] = result
    return float(result["metric"]), all_fused, info
