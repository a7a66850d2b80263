"""The N>1 path of bench.py (frame sharding + all_gather of results) on CPU: world_size 2,
gloo backend.  Checks that the sharded result equals the single-process result bit for bit, and the
stream-ordering contract of the gather (the gather of batch t overlaps the
compute of batch t+1): the collective waits for the batch's completion event only, on its own stream,
and nothing the compute pipeline waits on ever carries a collective."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_hot_path(frames):
    """Stand-in for the per-frame result [B,N,J,5]: any per-frame-independent function."""
    g = torch.Generator().manual_seed(0)
    w = torch.randn(5, generator=g)
    return (frames.view(-1, 1, 1, 1) * w.view(1, 1, 1, 5)).repeat(1, 10, 15, 1)


class _LogStream:
    """Records what is enqueued on / waited for by a stream (stands in for a HIP stream)."""

    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_event(self, ev):
        self.log.append((self.name, "wait_event", ev))

    def wait_stream(self, other):
        self.log.append((self.name, "wait_stream", other.name))

    def synchronize(self):
        self.log.append((self.name, "synchronize", None))


class _Ctx:
    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.stream.log.append((self.stream.name, "enter", None))

    def __exit__(self, *a):
        self.stream.log.append((self.stream.name, "exit", None))


class _MockPipeline:
    """PipelinedForward.submit's stream semantics: batch i runs on stream i % depth after waiting for
    the CALLER's stream (where the inputs were produced); returns (result, completion event)."""

    def __init__(self, depth, current, log):
        self.streams = [_LogStream(f"compute{k}", log) for k in range(depth)]
        self.current, self.log, self.i = current, log, 0

    def submit(self, frames):
        st = self.streams[self.i % len(self.streams)]
        st.wait_stream(self.current)
        ev = f"done{self.i}"
        self.log.append((st.name, "record", ev))
        self.i += 1
        return _fake_hot_path(frames), ev


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from faster_voxelpose_amd.core import distributed as D
    lo, hi = D.shard_frames(total, world, rank)
    log = []
    current = _LogStream("current", log)
    comm = _LogStream("comm", log)
    pipe = _MockPipeline(3, current, log)
    gat = D.ResultGatherer(world, stream=comm, stream_ctx=_Ctx)
    outs = []
    for step in range(4):                                   # 4 steps, 3 batches in flight
        local, ev = pipe.submit(torch.arange(lo, hi, dtype=torch.float32) + 100 * step)
        outs.append(gat.gather(local, ev).clone())
    gat.synchronize()
    dist.barrier()
    if rank == 0:
        q.put(([o.numpy() for o in outs], log))
    dist.destroy_process_group()


def test_sharded_gather_equals_single_process_and_never_fences_the_pipeline():
    from faster_voxelpose_amd.core import distributed as D
    total, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29611, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs, log = q.get()
    outs = [torch.from_numpy(o) for o in outs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for step, out in enumerate(outs):
        want = _fake_hot_path(torch.arange(0, total, dtype=torch.float32) + 100 * step)   # rank r's rows at [r*4,(r+1)*4)
        want = torch.cat([_fake_hot_path(torch.arange(r * 4, r * 4 + 4, dtype=torch.float32) + 100 * step)
                          for r in range(world)])
        assert torch.equal(out, want)
    # ordering contract
    comm_ops = [e for e in log if e[0] == "comm"]
    assert [e[2] for e in comm_ops if e[1] == "wait_event"] == ["done0", "done1", "done2", "done3"]
    assert not any(e[1] == "wait_stream" for e in comm_ops)
    compute_waits = [e for e in log if e[0].startswith("compute") and e[1] in ("wait_stream", "wait_event")]
    assert compute_waits and all(e[1] == "wait_stream" and e[2] == "current" for e in compute_waits), \
        "a compute stream may only wait for the submitting stream"
    assert not any(e[0] == "current" for e in log), "the submitting stream must carry neither collectives nor waits"
    # batch t+1 is submitted before gather t is synchronised: the only synchronize is the final one
    assert [e for e in log if e[1] == "synchronize"] == [("comm", "synchronize", None)]
    # world == 1: identity, no streams
    x = _fake_hot_path(torch.arange(0, 8, dtype=torch.float32))
    assert D.ResultGatherer(1).gather(x) is x
    assert D.shard_frames(8, 2, 1) == (4, 8)


def _worker_always(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from faster_voxelpose_amd.core import distributed as D
    log = []
    comm = _LogStream("comm", log)
    gat = D.ResultGatherer(1, stream=comm, stream_ctx=_Ctx, always=True)
    x = _fake_hot_path(torch.arange(0, 8, dtype=torch.float32))
    out = gat.gather(x, "done0")
    gat.synchronize()
    q.put((out.clone().numpy(), out.data_ptr() != x.data_ptr(), log))
    dist.destroy_process_group()


def test_forced_gather_at_world_size_one_runs_the_collective():
    """``always=True`` (bench.py's FVP_BENCH_FORCE_DIST): the whole N > 1 path - communication stream, event
    wait, all_gather into a ring buffer - runs with a single rank too."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker_always, args=(29617, q))
    p.start()
    out, copied, log = q.get()
    out = torch.from_numpy(out)
    p.join(60)
    assert p.exitcode == 0
    assert torch.equal(out, _fake_hot_path(torch.arange(0, 8, dtype=torch.float32))) and copied
    assert ("comm", "wait_event", "done0") in log


class _MockEvent:
    """An event with torch.cuda.Event's host-side interface: complete after `after` calls of query(), or once synchronize()
    has been called."""

    def __init__(self, name, after, log):
        self.name, self.left, self.log = name, after, log

    def query(self):
        self.left -= 1
        return self.left < 0

    def synchronize(self):
        self.log.append(("host", "event_synchronize", self.name))
        self.left = -1


def _worker_host_issue(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from faster_voxelpose_amd.core import distributed as D
    lo, hi = D.shard_frames(total, world, rank)
    log = []
    comm = _LogStream("comm", log)
    gat = D.ResultGatherer(world, stream=comm, stream_ctx=_Ctx)            # issue="host" is the default
    outs, issued_after = [], []
    for step in range(4):
        local = _fake_hot_path(torch.arange(lo, hi, dtype=torch.float32) + 100 * step)
        # batch `step` finishes two polls after it was submitted (both ranks alike: the collectives must pair up)
        outs.append(gat.gather(local, _MockEvent(f"done{step}", 2, log)))
        issued_after.append(sum(1 for e in log if e == ("comm", "enter", None)))
    pending_before_sync = len(gat._pending)
    gat.synchronize()
    dist.barrier()
    if rank == 0:
        q.put(([o.clone().numpy() for o in outs], log, issued_after, pending_before_sync))
    dist.destroy_process_group()


def test_host_issued_gather_has_no_gpu_side_wait_and_keeps_the_order():
    """Round 6 default: the collective of a batch is enqueued by the HOST once the batch's event reports completion
    (ResultGatherer.poll), so no stream ever carries a wait for it - a waiting packet in the communication stream's queue cost
    the compute streams 7-9 % on the MI355X (core/distributed.py).  Gathers are issued in submission order, only for
    finished batches; synchronize() host-waits for the rest and issues them; results equal the single-process ones."""
    total, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker_host_issue, args=(r, world, 29623, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs, log, issued_after, pending_before_sync = q.get()
    outs = [torch.from_numpy(o) for o in outs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for step, out in enumerate(outs):
        want = torch.cat([_fake_hot_path(torch.arange(r * 4, r * 4 + 4, dtype=torch.float32) + 100 * step) for r in range(world)])
        assert torch.equal(out, want)
    assert not any(e[1] in ("wait_event", "wait_stream") for e in log), "host-issued gathers put no wait into any stream"
    # a batch's collective is never enqueued by the gather() call that submitted it (its event is not complete yet), the
    # backlog drains in order, and whatever is still pending at the end is completed by synchronize()
    assert issued_after[0] == 0 and issued_after == sorted(issued_after) and issued_after[-1] < 4
    assert pending_before_sync == 4 - issued_after[-1] >= 1
    assert sum(1 for e in log if e == ("comm", "enter", None)) == 4
    waited = [e[2] for e in log if e[1] == "event_synchronize"]
    assert waited == [f"done{k}" for k in range(4 - pending_before_sync, 4)]
    assert log[-1] == ("comm", "synchronize", None)


def _worker_no_event(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from faster_voxelpose_amd.core import distributed as D
    log = []
    comm, current = _LogStream("comm", log), _LogStream("current", log)
    gat = D.ResultGatherer(1, stream=comm, stream_ctx=_Ctx, always=True, current_stream=lambda: current)
    x = _fake_hot_path(torch.arange(0, 8, dtype=torch.float32))
    out = gat.gather(x)                       # no completion event (plain forward / hipGraph replay)
    gat.synchronize()
    q.put((out.clone().numpy(), log))
    dist.destroy_process_group()


def test_gather_without_event_is_ordered_behind_the_callers_stream():
    """with ``ready=None`` the collective used to run on the communication stream with no
    dependency on the stream that produced ``local`` (bench.py --graph at world > 1).  It must wait for the
    caller's current stream - and still nothing may wait on the communication stream."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker_no_event, args=(29619, q))
    p.start()
    out, log = q.get()
    out = torch.from_numpy(out)
    p.join(60)
    assert p.exitcode == 0
    assert torch.equal(out, _fake_hot_path(torch.arange(0, 8, dtype=torch.float32)))
    assert ("comm", "wait_stream", "current") in log
    assert log.index(("comm", "wait_stream", "current")) < log.index(("comm", "enter", None))
    assert not any(e[0] == "current" for e in log)


def test_bench_launcher_starts_its_own_ranks():
    """``python bench.py --gpus 2`` without WORLD_SIZE: bench.py spawns two ranks itself (VERDICT round 2,
    missing #1).  FVP_BENCH_STUB=1 swaps the GPU step for a CPU stand-in over gloo; launcher, barrier-bracketed
    timing, max-over-ranks and the one-JSON-line contract are the real code."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["FVP_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--batch", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    # gloo prints a connection banner through C stdio; the contract is ONE JSON line, the last line of stdout
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]
    assert len(lines) == 1, r.stdout
    assert r.stdout.strip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["frames_per_gpu_per_step"] == 3
    assert abs(d["value"] - 2 * 3 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    # a mismatching WORLD_SIZE is a clean error, not an assertion
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert r2.returncode == 2 and "WORLD_SIZE=1" in r2.stderr and "Traceback" not in r2.stderr


def test_bench_launcher_without_enough_gpus_exits_cleanly():
    import subprocess
    import torch as _t
    if _t.cuda.is_available() and _t.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FVP_BENCH_STUB")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 3 and "needs 2 visible GPUs" in r.stderr and "Traceback" not in r.stderr
    assert r.stdout.strip() == ""


def _worker_real_path(rank, world, port, q):
    """One rank of the frame-sharded job with the REAL hot path (the unmodified kernels on the CPU emulator)."""
    import ctypes
    import subprocess
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("FVP_WINO_GENERIC", "1")     # (the emulated library is a -DFVP_DIAG=1 build, see tests/conftest.py)
    from faster_voxelpose_amd import netspec
    netspec.WINO_GENERIC = True
    for p_ in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cases import make_inputs, make_weights
    from faster_voxelpose_amd import _capi as capi
    from faster_voxelpose_amd.core import distributed as D
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    here = os.path.join(ROOT, "tests", "hipemu")
    if rank == 0:
        subprocess.run([os.path.join(here, "build_emu.sh")], check=True, capture_output=True)
    dist.barrier()
    lib = capi.bind(ctypes.CDLL(os.path.join(here, "libfvp_emu.so")))
    case = "tiny_u_b3_thr"                                   # 3 frames in the fixture: use frames 0..1 per rank below
    cfg, cams, seq, rt, heat, meta, _ = make_inputs(case)
    model = FV.FasterVoxelPoseNet(cfg, _lib=lib)
    model.load_state_dict(make_weights(case, model.state_dict()))
    frames = torch.cat([heat, heat.flip(0)])[:4]             # a global batch of 4 distinct frames
    lo, hi = D.shard_frames(frames.shape[0], world, rank)
    gat = D.ResultGatherer(world)
    with torch.no_grad():
        fused = model(meta={"seq": [seq] * (hi - lo)}, input_heatmaps=frames[lo:hi].contiguous(), cameras=cams,
                      resize_transform=rt)[0]
        out = gat.gather(fused).clone()
        gat.synchronize()
        whole = model(meta={"seq": [seq] * frames.shape[0]}, input_heatmaps=frames, cameras=cams, resize_transform=rt)[0] \
            if rank == 0 else None
    dist.barrier()
    if rank == 0:
        q.put((out.numpy(), whole.numpy()))
    dist.destroy_process_group()


def test_sharded_real_hot_path_equals_single_process_bit_for_bit():
    """SURVEY 8e on the real path: two gloo ranks each run the hot path (emulated kernels, miniature config) on their
    contiguous shard of a 4-frame batch and all_gather the fused poses; the result equals one process running all four
    frames, bit for bit (frames are independent units: nothing in the path mixes them)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker_real_path, args=(r, world, 29627, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, whole = (torch.from_numpy(a) for a in q.get())
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert out.shape == whole.shape and out.shape[0] == 4
    assert torch.equal(out, whole)
    assert bool((whole[..., 3] >= 0).any()), "the fixture must contain valid people"
