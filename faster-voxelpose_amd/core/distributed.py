"""Multi-GPU runner pieces: frames are independent units, one process per
GPU, every rank runs the full hot path on its own frames, and the only collective is one
``all_gather`` (RCCL over xGMI on GPUs, gloo in the CPU tests) of the ``fused_poses`` rows.

The gather must never fence the compute pipeline: several batches are in flight on their own HIP
streams (``PipelinedForward``), so the collective of batch t is issued on a dedicated communication
stream that waits for batch t's completion *event* only.  No compute stream ever waits on the
communication stream, and the stream that submits batches (the caller's current stream) carries no
collective, so batch t+1 starts while gather t is still running.
"""
import torch
import torch.distributed as dist


def shard_frames(total_frames, world, rank):
    """Contiguous ``total/world`` frames per rank: rank r owns ``[r*per, (r+1)*per)``."""
    per = total_frames // world
    return rank * per, (rank + 1) * per


class _InlineStream:
    """CPU stand-in for a HIP stream (gloo tests): work runs inline, events are always complete."""

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class ResultGatherer:
    """``gather(local, ready)`` -> ``[world * B, ...]`` tensor holding every rank's ``local`` rows in
    rank order.  On a GPU the collective runs on ``self.stream`` (a dedicated communication stream),
    ordered after ``ready`` (the event recorded when ``local`` was produced); the returned tensor is
    valid once ``synchronize()`` returns (or after ``self.stream`` in stream order).  With
    ``world == 1`` the input is returned untouched and no stream is involved.  Without ``ready`` the collective
    is ordered behind everything already enqueued on the caller's current stream.

    ``stream`` / ``stream_ctx`` are injectable so the ordering contract can be tested without a GPU."""

    def __init__(self, world, device=None, stream=None, stream_ctx=None, always=False, current_stream=None):
        self.world = int(world)
        self.always = bool(always)      # run the collective for world == 1 too (single-GPU check of the RCCL path)
        self.device = torch.device(device) if device is not None else None
        on_gpu = self.device is not None and self.device.type == "cuda"
        if stream is not None:
            self.stream = stream
        elif on_gpu and (self.world > 1 or self.always):
            self.stream = torch.cuda.Stream(device=self.device)
        else:
            self.stream = _InlineStream()
        self._ctx = stream_ctx if stream_ctx is not None else (torch.cuda.stream if on_gpu and (self.world > 1 or self.always) else None)
        # the stream a gather WITHOUT a completion event is ordered behind (the caller's current stream)
        if current_stream is not None:
            self._current = current_stream
        elif isinstance(self.stream, _InlineStream) or not on_gpu:
            self._current = None
        else:
            self._current = lambda: torch.cuda.current_stream(self.device)
        self._out = {}
        self._slot = 0

    def _buffer(self, local, slots=4):
        # a small ring of output buffers: results of the last `slots` gathers stay valid
        key = (tuple(local.shape), local.dtype, self._slot % slots)
        self._slot += 1
        buf = self._out.get(key)
        if buf is None:
            buf = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
            self._out[key] = buf
        return buf

    def gather(self, local, ready=None):
        if self.world == 1 and not self.always:
            return local
        if ready is not None:
            self.stream.wait_event(ready)               # the ONLY dependency: batch t -> gather t
        elif self._current is not None:
            # no completion event given: `local` was produced by work already enqueued on the caller's current
            # stream (plain forward, hipGraph replay), so order the collective behind that stream.  The caller
            # must not overwrite `local` before the gather has run (a graph's static output: pass a clone).
            self.stream.wait_stream(self._current())
        out = self._buffer(local)

        def run():
            if local.is_cuda:
                local.record_stream(self.stream)        # allocated on a pipeline stream, read here
            dist.all_gather_into_tensor(out, local.contiguous())

        if self._ctx is not None:
            with self._ctx(self.stream):
                run()
        else:
            run()
        return out

    def synchronize(self):
        self.stream.synchronize()
