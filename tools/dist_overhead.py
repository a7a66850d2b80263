"""Cost of the multi-GPU result gather on ONE GPU (diagnostics): RCCL process group of world size 1, three batches in
flight, the gather of every batch on its own stream (core/distributed.py) against no gather / a plain copy.
   gpurun -- python tools/dist_overhead.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
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
pipe = FV.PipelinedForward(model, depth=3)
comm = torch.cuda.Stream()
bufs = [torch.empty(B, 10, 15, 5, device=dev) for _ in range(4)]
def run(mode, steps=30):
    g = D.ResultGatherer(1, device=dev, always=(mode == "gather"))
    with torch.no_grad():
        for i in range(4):
            pipe.submit(meta=meta, cameras=cams, resize_transform=rt, input_heatmaps=heats[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tsub = 0.0
        for i in range(steps):
            (fused, _, _, _, _), ev = pipe.submit(meta=meta, cameras=cams, resize_transform=rt, input_heatmaps=heats[i % 4])
            ta = time.perf_counter()
            if mode == "gather":
                g.gather(fused, ev)
            elif mode == "copy":
                comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    fused.record_stream(comm); bufs[i % 4].copy_(fused)
            elif mode == "allgather_cur":
                dist.all_gather_into_tensor(bufs[i % 4], fused)
            tsub += time.perf_counter() - ta
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{mode:14s} fps {steps*B/(t2-t0):8.1f}  submit loop {1e3*(t1-t0)/steps:.3f} ms/step  gather call {1e3*tsub/steps:.3f} ms/step  drain {1e3*(t2-t1):.2f} ms")
for m, st in [("gather", 3), ("gather", 20), ("gather", 20), ("none", 20), ("gather", 60), ("none", 60)]:
    if os.environ.get("BARRIER"): dist.barrier()
    run(m, st)
dist.destroy_process_group()
