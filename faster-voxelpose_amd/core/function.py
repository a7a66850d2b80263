"""Validation loops over the hot path.

``validate(config, backbone, model, loader, output_dir, has_evaluate_function=False)`` is the drop-in for the
reference's ``validate`` (``lib/core/function.py:117-174``): same signature, same loader protocol (a
``torch.utils.data.DataLoader``-like iterable of ``(inputs, targets, meta, input_heatmaps)`` whose ``dataset`` carries
``cameras``, ``resize_transform`` and ``evaluate``), same return value - so ``run/validate.py:97-102`` calls it unchanged.

``validate_batches`` is the convenience form the tools / tests of this tree use: batches in, ``fused_poses`` out,
optionally scored.  A batch is a dict with ``meta`` (at least ``{'seq': [...]}``) and either ``input_heatmaps``
``[B,V,J,H,W]`` (the ``'image'`` / ``'gt'`` sources, heatmaps already computed) or
``pred_pose2d`` (the ``'pred'`` source of Shelf / Campus: per frame, per view, a list of
``[J,>=2]`` detections in original-image pixels), which is rasterised on the GPU
(``dataset.heatmaps.generate_input_heatmaps``).  With ``depth > 1`` batches are kept in flight
on separate HIP streams (``PipelinedForward``)."""
import logging
import os
import time

import torch

from ..dataset.heatmaps import generate_input_heatmaps
from ..models.faster_voxelpose import PipelinedForward

logger = logging.getLogger(__name__)


def validate_batches(config, model, batches, cameras, resize_transform, evaluate=None, depth=1, log=None):
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


class _Mean:
    """Running mean of a timing (the reference's AverageMeter, function.py:177-193: ``val`` = last, ``avg`` = mean)."""

    def __init__(self):
        self.val = self.sum = 0.0
        self.count = 0

    def update(self, v, n=1):
        self.val = v
        self.sum += v * n
        self.count += n

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0.0


def validate(config, backbone, model, loader, output_dir, has_evaluate_function=False):
    """Drop-in for ``lib/core/function.py:117-174``.  ``loader`` yields ``(inputs, targets, meta, input_heatmaps)``;
    ``loader.dataset`` supplies ``cameras``, ``resize_transform`` and - when ``has_evaluate_function`` -
    ``evaluate(all_fused_poses) -> (metric, message)``.  With ``config.DATASET.TEST_HEATMAP_SRC == 'image'`` the views go
    through ``backbone`` (:136-141), otherwise the loader's heatmaps are used (:142-148).  Returns the metric (0.0
    without an evaluate function, :168-169)."""
    model.eval()
    if backbone is not None:
        backbone.eval()
    cameras = loader.dataset.cameras
    dev = config.DEVICE
    resize_transform = torch.as_tensor(loader.dataset.resize_transform, dtype=torch.float, device=dev)
    from_images = config.DATASET.TEST_HEATMAP_SRC == "image"
    step_time, wait_time = _Mean(), _Mean()
    collected = []
    nbatches = len(loader) if hasattr(loader, "__len__") else None
    with torch.no_grad():
        mark = time.time()
        for i, (inputs, _, meta, input_heatmaps) in enumerate(loader):
            wait_time.update(time.time() - mark)
            if from_images:
                inputs = inputs.to(dev)
                fused_poses, plane_poses, proposal_centers, input_heatmaps, _ = model(
                    backbone=backbone, views=inputs, meta=meta, cameras=cameras, resize_transform=resize_transform)
            else:
                input_heatmaps = input_heatmaps.to(dev)
                fused_poses, plane_poses, proposal_centers, _, _ = model(
                    backbone=backbone, meta=meta, input_heatmaps=input_heatmaps, cameras=cameras,
                    resize_transform=resize_transform)
            collected.append(fused_poses)
            step_time.update(time.time() - mark)
            mark = time.time()
            if i % config.PRINT_FREQ == 0 or (nbatches is not None and i == nbatches - 1):
                nsamples = fused_poses.shape[0] * (inputs.shape[1] if torch.is_tensor(inputs) and inputs.dim() > 1
                                                   else input_heatmaps.shape[1])
                mem = torch.cuda.memory_allocated(0) if torch.cuda.is_available() else 0
                logger.info("Test: [%d/%s]\tTime: %.3fs (%.3fs)\tSpeed: %.1f samples/s\tData: %.3fs (%.3fs)\tMemory %.1f",
                            i, nbatches if nbatches is not None else "?", step_time.val, step_time.avg,
                            nsamples / max(step_time.val, 1e-9), wait_time.val, wait_time.avg, mem)
                if config.TEST.VISUALIZATION:
                    from ..utils.vis import test_vis_all
                    prefix = "{}_{:08}".format(os.path.join(output_dir, "validation"), i)
                    test_vis_all(config, meta, cameras, resize_transform, inputs, input_heatmaps, fused_poses, plane_poses,
                                 proposal_centers, prefix)
        all_fused_poses = torch.cat(collected, dim=0)
    if not has_evaluate_function:
        return 0.0
    metric, msg = loader.dataset.evaluate(all_fused_poses)
    logger.info(msg)
    return metric
