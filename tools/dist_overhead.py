"""Cost of the multi-GPU result gather on ONE GPU (diagnostics): RCCL process group of world size 1, three batches in
flight, the gather of every batch on its own stream (core/distributed.py) against no gather / a plain copy.
   gpurun -- python tools/dist_overhead.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")     # as bench.py: the compute streams, the gather stream and the default stream on their own queues
import torch, torch.distributed as dist
import fvp_synthetic as S
from faster_voxelpose_amd.core import distributed as D
from faster_voxelpose_amd.models import faster_voxelpose as FV
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = "cuda:0"
cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
cams, seq = S.load_cameras("panoptic"); rt = S.resize_transform(cfg).to(dev)
B = 8
heats = [S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100 + i).to(dev) for i in range(4)]
meta = {"seq": [seq] * B}
model = FV.get(cfg).to(dev); model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
pipe = FV.PipelinedForward(model, depth=4)
comm = torch.cuda.Stream()
bufs = [torch.empty(B, 10, 15, 5, device=dev) for _ in range(4)]
stage = torch.empty(4, B, 10, 15, 5, device=dev)
keep = []
fifo = []
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def mk(flags):
    out = []
    for _ in range(8):
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(flags)) == 0
        out.append(e)
    return out
raw_nf = mk(0x2 | 0x20000000)      # hipEventDisableTiming | hipEventDisableSystemFence
raw_df = mk(0x2)                   # hipEventDisableTiming
flags = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(4)]
big = torch.empty(4, B, 10, 15, 5, device=dev)
_orig_wait_stream = torch.cuda.Stream.wait_stream
def run(mode, steps=30):
    # "nowait_*": PipelinedForward.submit without its st.wait_stream(current stream) (the inputs are resident here): does the
    # per-batch cross-stream wait on the IDLE submitting stream cost anything?
    torch.cuda.Stream.wait_stream = (lambda self, other: None) if mode.startswith("nowait") else _orig_wait_stream
    g = D.ResultGatherer(1, device=dev, always=(mode == "gather"))
    fifo.clear()
    issued_in_loop = 0
    for f_ in flags: f_.zero_()
    torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(4):
            pipe.submit(meta=meta, cameras=cams, resize_transform=rt, input_heatmaps=heats[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tsub = 0.0
        evs = []
        for i in range(steps):
            if "throttle" in mode and i >= 4:
                evs[i - 4].synchronize()                # host back-pressure: at most `depth` batches ahead of the GPU
            (fused, _, _, _, _), ev = pipe.submit(meta=meta, cameras=cams, resize_transform=rt, input_heatmaps=heats[i % 4])
            evs.append(ev)
            ta = time.perf_counter()
            if mode == "gather":
                g.gather(fused, ev)
            elif mode == "copy":
                comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    fused.record_stream(comm); bufs[i % 4].copy_(fused)
            elif mode in ("wait_nofence", "wait_rawdefault"):   # the dependency through a raw HIP event: without / with the system-scope fence
                e = raw_nf[i % 8] if mode == "wait_nofence" else raw_df[i % 8]
                st = pipe.streams[i % 4]
                assert hip.hipEventRecord(e, C.c_void_p(st.cuda_stream)) == 0
                assert hip.hipStreamWaitEvent(C.c_void_p(comm.cuda_stream), e, 0) == 0
            elif mode in ("query_copy", "query_gather", "throttle_gather", "throttle_copy"):    # no GPU-side dependency: the host issues the transfer once event.query() says done
                fifo.append((ev, fused, i))
                while fifo and fifo[0][0].query():
                    _, f, k = fifo.pop(0)
                    with torch.cuda.stream(comm):
                        if mode in ("query_copy", "throttle_copy"):
                            bufs[k % 4].copy_(f)
                        else:
                            dist.all_gather_into_tensor(bufs[k % 4], f)
            elif mode in ("memop_wait", "memop_copy", "memop_gather"):   # dependency through stream memory operations instead of an event
                st = pipe.streams[i % 4]
                flag = flags[i % 4]
                assert hip.hipStreamWriteValue32(C.c_void_p(st.cuda_stream), C.c_void_p(flag.data_ptr()), C.c_uint32(i + 1), C.c_uint(0)) == 0
                assert hip.hipStreamWaitValue32(C.c_void_p(comm.cuda_stream), C.c_void_p(flag.data_ptr()), C.c_uint32(i + 1), C.c_uint(0), C.c_uint32(0xffffffff)) == 0
                if mode != "memop_wait":
                    with torch.cuda.stream(comm):
                        if mode == "memop_copy":
                            bufs[i % 4].copy_(fused)
                        else:
                            dist.all_gather_into_tensor(bufs[i % 4], fused)
                    keep.append(fused)
                    if len(keep) > 8: keep.pop(0)
            elif mode == "wait_only":                    # the cross-stream dependency alone
                comm.wait_event(ev)
            elif mode == "copy_norecord":                # without telling the caching allocator
                comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    bufs[i % 4].copy_(fused)
                keep.append(fused)
                if len(keep) > 8: keep.pop(0)
            elif mode == "copy_samestream":              # the copy on the batch's own stream: no cross-stream anything
                with torch.cuda.stream(pipe.streams[i % 4]):
                    bufs[i % 4].copy_(fused)
            elif mode == "allgather_cur":
                dist.all_gather_into_tensor(bufs[i % 4], fused)
            elif mode == "gather4":                      # a staging copy per batch, ONE collective per four batches
                comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    fused.record_stream(comm); stage[i % 4].copy_(fused)
                    if i % 4 == 3:
                        dist.all_gather_into_tensor(big, stage)
            tsub += time.perf_counter() - ta
        t1 = time.perf_counter()
        issued_in_loop = steps - len(fifo)
        while fifo:                                      # flush: the rest is issued as the batches complete
            e_, f, k = fifo.pop(0)
            e_.synchronize()
            with torch.cuda.stream(comm):
                if mode in ("query_copy", "throttle_copy"):
                    bufs[k % 4].copy_(f)
                else:
                    dist.all_gather_into_tensor(bufs[k % 4], f)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{mode:14s} fps {steps*B/(t2-t0):8.1f}  submit loop {1e3*(t1-t0)/steps:.3f} ms/step  gather call {1e3*tsub/steps:.3f} ms/step  drain {1e3*(t2-t1):.2f} ms  issued in loop {issued_in_loop}")
for m, st in [("gather", 3), ("none", 100), ("nowait_none", 100), ("throttle_none", 100), ("nowait_throttle", 100), ("none", 100), ("nowait_none", 100), ("throttle_none", 100), ("nowait_throttle", 100)]:
    if os.environ.get("BARRIER"): dist.barrier()
    run(m, st)
dist.destroy_process_group()
