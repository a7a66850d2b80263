"""Top module -- drop-in for the reference's ``lib/models/faster_voxelpose.py``
(``FasterVoxelPoseNet`` :18-105, ``get`` :108), inference branch.

``forward(backbone=None, views=None, meta=None, targets=None, input_heatmaps=None,
cameras=None, resize_transform=None)`` returns
``(fused_poses [B,N,J,5], plane_poses [3,B,N,J,2], proposal_centers [B,N,7], input_heatmaps, None)``.
"""
import time

import torch
import torch.nn as nn

from ..engine import HotPath
from .human_detection_net import HumanDetectionNet
from .joint_localization_net import JointLocalizationNet


class FasterVoxelPoseNet(nn.Module):
    def __init__(self, cfg, _lib=None):
        super().__init__()
        self.cfg = cfg
        self.max_people = cfg.CAPTURE_SPEC.MAX_PEOPLE
        self.num_joints = cfg.DATASET.NUM_JOINTS
        self.device = torch.device(cfg.DEVICE)
        self.engine = HotPath(cfg, _lib=_lib)
        self.pose_net = HumanDetectionNet(cfg, _engine=self.engine)
        self.joint_net = JointLocalizationNet(cfg, _engine=self.engine)
        self.lambda_loss_2d = cfg.TRAIN.LAMBDA_LOSS_2D
        self.lambda_loss_1d = cfg.TRAIN.LAMBDA_LOSS_1D
        self.lambda_loss_bbox = cfg.TRAIN.LAMBDA_LOSS_BBOX
        self.lambda_loss_fused = cfg.TRAIN.LAMBDA_LOSS_FUSED
        self.eval()

    def forward(self, backbone=None, views=None, meta=None, targets=None, input_heatmaps=None, cameras=None,
                resize_transform=None):
        if self.training:
            raise NotImplementedError("only the inference branch of FasterVoxelPoseNet.forward is implemented "
                                      "(call model.eval()); training losses are outside the hot path")
        if views is not None:
            if hasattr(backbone, "_run"):
                # bf16 HIP backbone (models/resnet.py): all B*V views in one pass; it writes the heatmaps as
                # NCHW (returned, like the reference) and in the channels-last layout the projection reads
                B, V = views.shape[:2]
                nchw, cl = backbone._run(views.flatten(0, 1), True, True)
                input_heatmaps = nchw.view(B, V, *nchw.shape[1:])
                if cl.shape[-1] == self.engine.JP:
                    self.engine.adopt_staging(input_heatmaps, cl)
            else:
                # per-view backbone passes, as the reference (:36-38); any torch module
                num_views = views.shape[1]
                input_heatmaps = torch.stack([backbone(views[:, c]) for c in range(num_views)], dim=1)
        _, _, proposal_centers, _ = self.pose_net(input_heatmaps, meta, cameras, resize_transform)
        mask = self.engine.last["valid"]              # uint8 [B,N] = proposal_centers[:, :, 3] >= 0 (:45), written by fvp_proposals
        fused_poses, plane_poses = self.joint_net.forward5(meta, input_heatmaps, proposal_centers, mask, cameras,
                                                           resize_transform, _reuse_staging=True)
        # the channels-last staging copy is valid for this call only: a later tensor may reuse the
        # same address / version / shape once the caching allocator recycles the block
        self.engine.invalidate_staging()
        return fused_poses, plane_poses, proposal_centers, input_heatmaps, None


class GraphedForward:
    """The whole hot path captured once into a hipGraph and replayed per batch.

    The path has no host synchronisation and only static-shape launches, so one capture covers
    staging, HDN, JLN and fusion (~110 kernels); a replay costs one host call instead of ~110
    ctypes launches.  Inputs are copied into a static buffer; outputs are the graph's static
    tensors (clone them if they must outlive the next replay)."""

    def __init__(self, model, meta, input_heatmaps, cameras, resize_transform, warmup=2):
        self.model = model
        self.static_in = input_heatmaps.clone()
        self.args = dict(meta=meta, cameras=cameras, resize_transform=resize_transform)
        side = torch.cuda.Stream(device=input_heatmaps.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # packs weights, fills caches, sizes scratch
                model(input_heatmaps=self.static_in, **self.args)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.current_stream().synchronize()
        geo = model.engine.geo
        geo.pending = [ev for ev in geo.pending if not ev.query()]     # nothing left to query inside the capture
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = model(input_heatmaps=self.static_in, **self.args)
        self._pinned = (geo.cams, geo.fine_grid)        # the graph reads these through raw pointers: keep them alive

    def __call__(self, input_heatmaps=None):
        if input_heatmaps is not None and input_heatmaps.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(input_heatmaps, non_blocking=True)
        self.graph.replay()
        return self.out


class PipelinedForward:
    """Several batches in flight on one GPU: batch i runs on HIP stream i % depth with its own
    replica of the model (same weights, private scratch buffers).  The detection stage of a batch
    (CenterNet / C2CNet / NMS: small, latency-bound launches) then fills the gaps and tails of the
    previous batch's joint stage (P2PNet: large, matrix-core-bound launches); the forward has no
    host synchronisation, so submitting is just enqueueing.

    ``submit`` returns ``(outputs, event)``; the outputs are valid for a consumer stream after
    ``event.wait()`` (or after ``synchronize()``).  Results are identical to the plain forward:
    every kernel is deterministic and replicas share nothing but read-only inputs."""

    def __init__(self, model, depth=2, streams=None, backpressure=True):
        """``streams``: optional list of >= depth ``torch.cuda.Stream`` to run on (a process that builds several pipelines
        should hand them the same streams: the HIP runtime maps streams onto a fixed number of hardware queues -
        GPU_MAX_HW_QUEUES, default 4 - and streams that share a queue serialise).
        ``backpressure`` (round 6): before a slot is reused the HOST waits for the batch that ran in it last, so the host is
        never more than ``depth`` batches ahead of the GPU - nothing changes for the GPU (it always has depth - 1 batches
        queued: 3 235-3 239 frames/s with, 3 238-3 244 without), results can be consumed as they complete
        (``ResultGatherer.poll``) and the outputs of unboundedly many queued batches do not pile up.  ``wait_s`` accumulates
        the host time spent in that wait."""
        assert depth >= 1
        assert streams is None or len(streams) >= depth
        self.models = [model]
        self.streams = list(streams[:depth]) if streams is not None else [torch.cuda.Stream(device=model.device) for _ in range(depth)]
        for _ in range(1, depth):
            m = FasterVoxelPoseNet(model.cfg).to(model.device)
            m.load_state_dict(model.state_dict())
            m.eval()
            # camera tables and the per-sequence coordinate cache (164 MB for the Panoptic shape set) are read-only
            # between rebuilds: one copy for all replicas (engine.SharedGeometry orders the streams behind a rebuild)
            m.engine.geo = model.engine.geo
            self.models.append(m)
        self.depth = depth
        self._i = 0
        self.backpressure = bool(backpressure)
        self._last = [None] * depth                      # completion event of the batch that last ran in each slot
        self.wait_s = 0.0

    def submit(self, **forward_kwargs):
        k = self._i % self.depth
        self._i += 1
        st = self.streams[k]
        if self.backpressure and self._last[k] is not None:
            t0 = time.perf_counter()
            self._last[k].synchronize()
            self.wait_s += time.perf_counter() - t0
        st.wait_stream(torch.cuda.current_stream())      # inputs produced on the caller's stream
        # the inputs were allocated on the caller's stream but are read on `st`: tell the caching
        # allocator, or a caller that drops them right after submit() could see the block handed
        # out again (and overwritten on its own stream) while this batch is still queued on `st`
        for v in forward_kwargs.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(st)
        with torch.cuda.stream(st), torch.no_grad():
            out = self.models[k](**forward_kwargs)
            ev = torch.cuda.Event()
            ev.record(st)
        self._last[k] = ev
        return out, ev

    @staticmethod
    def consume(outputs, stream=None):
        """Declare that ``outputs`` (allocated on a pipeline stream) are about to be read on
        ``stream`` (default: the current one) so their memory is not recycled under the reader."""
        stream = stream or torch.cuda.current_stream()
        for t in outputs:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)

    def synchronize(self):
        for st in self.streams:
            st.synchronize()


class GraphedPipeline:
    """``PipelinedForward`` with every slot captured as a hipGraph (round 5): slot k = replica k + HIP stream k + one graph
    of the whole forward on that stream with a static input buffer.  ``submit(input_heatmaps)`` copies the batch into the
    slot's buffer and replays - ONE host call per batch instead of ~85 ctypes launches - and returns ``(outputs, event)``;
    the outputs are the slot's static tensors, overwritten when the slot comes round again (``depth`` submits later): read
    or clone them before that.  Same kernels, same order per slot: results equal ``PipelinedForward`` bit for bit.
    Shapes, cameras and the sequence list are fixed at capture time (the reference's caller loop, function.py:136-148,
    feeds one sequence mix per run); a different shape needs a new pipeline."""

    def __init__(self, model, depth, meta, input_heatmaps, cameras, resize_transform, streams=None, warmup=2):
        self.pipe = PipelinedForward(model, depth=depth, streams=streams)
        self.depth, self._i = depth, 0
        args = dict(meta=meta, cameras=cameras, resize_transform=resize_transform)
        self.static_in, self.graphs, self.outs, self._pinned = [], [], [], []
        cur = torch.cuda.current_stream()
        for k in range(depth):
            st, m = self.pipe.streams[k], self.pipe.models[k]
            buf = input_heatmaps.clone()
            st.wait_stream(cur)
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(warmup):                  # packs weights, fills the shared caches, sizes the scratch
                    m(input_heatmaps=buf, **args)
            st.synchronize()                             # pending geometry rebuilds have completed ...
            # ... so their events are dropped BEFORE the capture: Event.query() inside a capture is legal on HIP today but
            # not under CUDA's global capture mode - _await_geometry then has nothing to query
            geo = m.engine.geo
            geo.pending = [ev for ev in geo.pending if not ev.query()]
            assert not geo.pending, "a geometry rebuild is still running on another stream: synchronise before capturing"
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st), torch.no_grad():
                out = m(input_heatmaps=buf, **args)
            # the graph holds RAW pointers to the shared camera table / coordinate cache: keep those tensors alive for the
            # graph's lifetime, whatever an eager run with a new sequence retires later
            self._pinned.append((geo.cams, geo.fine_grid))
            self.static_in.append(buf)
            self.graphs.append(g)
            self.outs.append(out)
        torch.cuda.synchronize()

    def submit(self, input_heatmaps):
        k = self._i % self.depth
        self._i += 1
        st = self.pipe.streams[k]
        st.wait_stream(torch.cuda.current_stream())
        input_heatmaps.record_stream(st)
        with torch.cuda.stream(st):
            self.static_in[k].copy_(input_heatmaps, non_blocking=True)
            self.graphs[k].replay()
            ev = torch.cuda.Event()
            ev.record(st)
        return self.outs[k], ev

    def synchronize(self):
        self.pipe.synchronize()


def get(cfg):
    return FasterVoxelPoseNet(cfg)
