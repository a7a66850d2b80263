"""Multi-GPU runner pieces: frames are independent units, one process per
GPU, every rank runs the full hot path on its own frames, and the only collective is one
``all_gather`` (RCCL over xGMI on GPUs, gloo in the CPU tests) of the ``fused_poses`` rows.

The gather must never fence the compute pipeline: several batches are in flight on their own HIP
streams (``PipelinedForward``), so the collective of batch t runs on a dedicated communication
stream, ordered behind batch t's completion *event* only.  No compute stream ever waits on the
communication stream, and the stream that submits batches (the caller's current stream) carries no
collective, so batch t+1 starts while gather t is still running.

Round 6: HOW the collective is ordered behind the event matters on this part.  A GPU-side wait
(``stream.wait_event``: a blocking packet at the head of the communication stream's hardware queue)
costs the compute streams 7-9 % of their throughput while it waits - measured at world size 1 with
the wait ALONE, no copy and no collective behind it, with HIP events with or without the system fence
and with ``hipStreamWaitValue32`` alike (``tools/dist_overhead.py``: 3 240 -> 2 990-3 040 frames/s;
the same copy issued on the batch's own stream costs nothing).  So the default is host-issued: the
gather of batch t is enqueued by the host once ``event.query()`` says the batch has finished
(``poll()``, called by every ``gather`` and drained by ``synchronize``), with no wait packet in any
queue; ``PipelinedForward``'s back-pressure keeps the host at most ``depth`` batches ahead of the
GPU, so the gathers still stream one to two batches behind the compute (96 of 100 issued inside the
submit loop) at 3 225-3 238 frames/s against 3 235-3 244 without any gather.
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
    valid once ``synchronize()`` returns.  With
    ``world == 1`` the input is returned untouched and no stream is involved.  Without ``ready`` the collective
    is ordered behind everything already enqueued on the caller's current stream.

    ``stream`` / ``stream_ctx`` are injectable so the ordering contract can be tested without a GPU."""

    def __init__(self, world, device=None, stream=None, stream_ctx=None, always=False, current_stream=None, issue="host"):
        """``issue``: "host" (default) - the host enqueues the collective once the batch's event reports completion (needs
        events with ``query()`` / ``synchronize()``: torch.cuda.Event; anything else falls back to "stream"); "stream" - the
        communication stream waits for the event on the GPU (the pre-round-6 behaviour, 7-9 % slower on the MI355X)."""
        assert issue in ("host", "stream")
        self.issue = issue
        self._pending = []              # host-issued gathers waiting for their batch: (event, local, out), FIFO
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

    def _buffer(self, local, slots=8):
        # a small ring of output buffers: results of the last `slots` gathers stay valid
        key = (tuple(local.shape), local.dtype, self._slot % slots)
        self._slot += 1
        buf = self._out.get(key)
        if buf is None:
            buf = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
            self._out[key] = buf
        return buf

    def _issue(self, local, out):
        def run():
            if local.is_cuda:
                local.record_stream(self.stream)        # allocated on a pipeline stream, read here
            dist.all_gather_into_tensor(out, local.contiguous())

        if self._ctx is not None:
            with self._ctx(self.stream):
                run()
        else:
            run()

    def poll(self):
        """Enqueue the collectives of every pending batch that has finished, in submission order (all ranks issue their
        collectives in the same order: a batch is never gathered before its predecessors)."""
        while self._pending and self._pending[0][0].query():
            _, local, out = self._pending.pop(0)
            self._issue(local, out)

    def gather(self, local, ready=None):
        if self.world == 1 and not self.always:
            return local
        if self.issue == "host" and (hasattr(ready, "query") if ready is not None else local.is_cuda):
            # host-issued (module docstring): nothing waits on the GPU; the result is valid after synchronize()
            if ready is None:                           # plain forward / hipGraph replay on the caller's stream
                ready = torch.cuda.Event()
                ready.record(self._current() if self._current is not None else torch.cuda.current_stream(local.device))
            out = self._buffer(local)
            self._pending.append((ready, local, out))
            self.poll()
            return out
        if ready is not None:
            self.stream.wait_event(ready)               # the ONLY dependency: batch t -> gather t
        elif self._current is not None:
            # no completion event given: `local` was produced by work already enqueued on the caller's current
            # stream (plain forward, hipGraph replay), so order the collective behind that stream.  The caller
            # must not overwrite `local` before the gather has run (a graph's static output: pass a clone).
            self.stream.wait_stream(self._current())
        out = self._buffer(local)
        self._issue(local, out)
        return out

    def synchronize(self):
        while self._pending:                             # the rest is issued as the batches complete
            ev, local, out = self._pending.pop(0)
            ev.synchronize()
            self._issue(local, out)
        self.stream.synchronize()
