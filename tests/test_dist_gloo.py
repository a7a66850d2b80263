"""The N>1 path of bench.py (frame sharding + all_gather of results) on CPU: world_size 2,
gloo backend.  Checks the sharded result equals the single-process result bit for bit."""
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


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    lo, hi = bench.shard_frames(total, world, rank)
    local = _fake_hot_path(torch.arange(lo, hi, dtype=torch.float32))
    out = bench.gather_results(local, world)
    dist.barrier()
    if rank == 0:
        q.put(out)
    dist.destroy_process_group()


def test_sharded_gather_equals_single_process():
    import bench
    total, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29611, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = _fake_hot_path(torch.arange(0, total, dtype=torch.float32))
    assert torch.equal(out, want)
    assert bench.shard_frames(8, 2, 1) == (4, 8)
    assert torch.equal(bench.gather_results(want, 1), want)
