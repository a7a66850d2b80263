#!/usr/bin/env python
"""Throughput of the heatmaps -> 3-D joints hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A step = one pass of the whole hot path (staging, HDN, JLN, fusion) over one batch of B
synthetic frames per GPU: Panoptic shape set (5 views, 15 joints, 240x128 heatmaps, 80x80x20
voxels, jln64, MAX_PEOPLE 10), Gaussian-blob heatmaps resident in HBM, seeded random weights,
MIN_SCORE = -1 so that all 10 proposals per frame are valid (P = 10, the FLOP count
BASELINE.md quotes).  Frames shard across ranks (one process per GPU, weak scaling); the only
collective is an all_gather of the [B,10,15,5] result.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA peak


def algorithmic_bytes_projection(V, J, H, W, C, people_per_frame):
    """fused project_individual -> tri-plane: heatmaps read once per frame + 3 planes written per
    person: 4*V*J*H*W + P*3*4*J*C*C bytes per frame."""
    return 4.0 * V * J * H * W + people_per_frame * 3 * 4.0 * J * C * C


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo; falls back to half the logical CPUs."""
    try:
        seen, pkg = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pkg = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    seen.add((pkg, line.split(":")[1].strip()))
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name, seed, budget_s=40.0):
    """The CPU oracle (a port of the reference's PyTorch path, pinned against the reference's
    golden vectors) on the host cores of this box, same workload (P = 10 valid people per frame).
    Protocol of SURVEY.md section 8d, bounded to ~``budget_s`` of CPU work: threads = 8 and all
    physical cores, B = 1 and B = 8, 2 warm-ups, sampling grid pre-built, per-stage ms; the iteration
    counts are scaled so the default bench run stays within minutes (10 at B = 1, 2 at B = 8)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fvp_oracle as O
    import fvp_synthetic as S
    cfg = S.make_cfg(cfg_name, device="cpu", min_score=-1.0)
    cams, seq = S.load_cameras(cfg_name)
    rt = S.resize_transform(cfg)
    heat8 = S.heatmaps_blobs(cfg, cams, seq, 8, people=4, seed=seed)
    orc = O.Oracle(cfg, S.fill_state_dict(O.reference_state_dict_shapes(cfg), seed=7))
    phys = physical_cores()
    runs = []
    t_start = time.perf_counter()
    for threads in sorted({8, phys}):
        torch.set_num_threads(threads)
        for B, iters in ((1, 10), (8, 2)):
            heat = heat8[:B]
            meta = {"seq": [seq] * B}
            stages = {"project_whole+center_net+nms+c2c (hdn)": 0.0, "project_individual+p2p+softargmax+fusion (jln)": 0.0}
            with torch.no_grad():
                for it in range(-2, iters):                 # 2 warm-ups (build the sampling grid)
                    if it == 0:
                        t0 = time.perf_counter()
                        for k in stages:
                            stages[k] = 0.0
                    ta = time.perf_counter()
                    _, _, centers, _ = orc.hdn(heat, meta, cams, rt)
                    tb = time.perf_counter()
                    mask = centers[:, :, 3] >= 0
                    orc.jln(meta, heat, centers, mask, cams, rt)
                    tc = time.perf_counter()
                    stages["project_whole+center_net+nms+c2c (hdn)"] += tb - ta
                    stages["project_individual+p2p+softargmax+fusion (jln)"] += tc - tb
                    if it >= 0 and time.perf_counter() - t_start > budget_s and it >= 1:
                        iters = it + 1
                        break
            dt = time.perf_counter() - t0
            runs.append({"threads": threads, "batch": B, "iterations": iters, "frames_per_s": B * iters / dt,
                         "ms_per_frame_by_stage": {k: 1e3 * v / (B * iters) for k, v in stages.items()}})
    best = max(runs, key=lambda r: r["frames_per_s"])
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
            "sample": f"torch-CPU oracle on {cpu_model()} ({phys} physical cores, {os.cpu_count()} logical): threads 8 and "
                      f"{phys}, B = 1 (10 iterations) and B = 8 (2 iterations) after 2 warm-ups, P = 10 people/frame; "
                      f"value = best of the runs below", "runs": runs}


def mpjpe_vs_reference(dev):
    """BASELINE.json's second metric half: the committed float-parity fixture (outputs of the REFERENCE
    itself, tests/golden/make_golden.py) replayed through the HIP path; mean / max joint distance in mm."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import make_inputs, make_weights
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    out = {}
    for case in ("panoptic_c_b2_thr", "shelf_c_b1_thr"):
        cfg, cams, seq, rt, heat, meta, _ = make_inputs(case, device=dev)
        g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
        model = FV.get(cfg).to(dev)
        model.load_state_dict(make_weights(case, model.state_dict()))
        with torch.no_grad():
            fused = model(meta=meta, input_heatmaps=heat.to(dev), cameras=cams, resize_transform=rt.to(dev))[0]
        v = g["valid"]
        d = np.linalg.norm((fused[..., :3].cpu().numpy() - g["fused_poses"][..., :3])[v], axis=-1)
        out[case] = {"mean": float(d.mean()), "max": float(d.max()), "joints": int(d.size),
                     "reference_fp32_vs_fp64_floor_max": float(g["margins"][5])}
    first = out["panoptic_c_b2_thr"]
    return {"fixture": "panoptic_c_b2_thr (Panoptic 5-view 80x80x20 jln64, 8 valid people; reference outputs committed "
                       "under tests/golden)", "mean": first["mean"], "max": first["max"], "bar": 1e-3, "all": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--config", default="panoptic")
    ap.add_argument("--cpu-budget", type=float, default=40.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--inputs", type=int, default=4, help="distinct resident input batches rotated through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight: step i runs on HIP stream i %% S with its own scratch buffers, so the "
                         "detection stage of one batch overlaps the joint stage of the previous one")
    ap.add_argument("--prof-steps", type=int, default=5,
                    help="extra single-stream steps after the timed region with the per-class HIP-event timers on")
    ap.add_argument("--backbone", action="store_true",
                    help="end-to-end variant (BASELINE configs[4] shape): the step starts from 5 x [3,512,960] "
                         "images per frame and runs the bf16 Pose-ResNet-50 backbone in front of the voxel path")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured hipGraph (per-kernel event timing is then unavailable)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    # FVP_BENCH_FORCE_DIST=1: initialise RCCL and run the result gather for a single rank too (a one-GPU check of
    # the N > 1 code path: process group, communication stream, all_gather_into_tensor)
    force_dist = world == 1 and os.environ.get("FVP_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import fvp_synthetic as S
    from faster_voxelpose_amd import _capi as capi
    from faster_voxelpose_amd.core import distributed as D
    from faster_voxelpose_amd.models import faster_voxelpose as FV

    cfg = S.make_cfg(args.config, device=dev, min_score=-1.0)
    cams, seq = S.load_cameras(args.config)
    rt = S.resize_transform(cfg).to(dev)
    B = args.batch
    # weak scaling: every rank owns B frames per step (rank r = frames [r*B, (r+1)*B) of the global batch);
    # `--inputs` distinct batches per rank stay resident in HBM and are rotated through the steps
    lo, hi = D.shard_frames(B * world, world, rank)
    assert hi - lo == B
    nin = max(1, args.inputs)
    heats = [S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100 + 16 * rank + i).to(dev) for i in range(nin)]
    heat = heats[0]
    meta = {"seq": [seq] * B}
    model = FV.get(cfg).to(dev)
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
    lib = capi.load()
    # batches in flight: FV.PipelinedForward = one replica (scratch buffers) + one HIP stream each
    nstreams = max(1, args.streams)
    pipe = FV.PipelinedForward(model, depth=nstreams) if nstreams > 1 else None

    views = bb = None
    if args.backbone:
        from faster_voxelpose_amd.core import config as CFG
        from faster_voxelpose_amd.models import resnet as RN
        bb = RN.get(CFG.default_config()).to(dev)
        bb.load_state_dict(S.fill_backbone_state_dict(bb.state_dict(), seed=3))
        Wi, Hi = cfg.DATASET.IMAGE_SIZE
        views_all = [torch.rand(B, cfg.DATASET.CAMERA_NUM, 3, Hi, Wi, device=dev) for _ in range(min(nin, 2))]
        views = views_all[0]
        args.no_prof = True

    graphed = None
    if args.graph:
        args.no_prof = True
        graphed = FV.GraphedForward(model, meta, heat, cams, rt)

    # the gather of batch t runs on its own stream behind batch t's completion event, so it never fences
    # the compute pipeline (core/distributed.py)
    gatherer = D.ResultGatherer(world, device=dev, always=force_dist)

    def step(i=0, pipelined=True):
        if bb is not None:
            kw = dict(backbone=bb, views=views_all[i % len(views_all)])
        else:
            kw = dict(input_heatmaps=heats[i % nin])
        if graphed is not None:
            return gatherer.gather(graphed(heats[i % nin])[0])
        if pipe is not None and pipelined:
            (fused, planes, centers, _, _), ev = pipe.submit(meta=meta, cameras=cams, resize_transform=rt, **kw)
            return gatherer.gather(fused, ev)
        fused, planes, centers, _, _ = model(meta=meta, cameras=cams, resize_transform=rt, **kw)
        ev = torch.cuda.Event()
        ev.record()
        return gatherer.gather(fused, ev)

    if world > 1 or force_dist:
        # RCCL builds its communicator lazily inside the first collective (seconds of host time with an idle GPU);
        # do that before the warm-up steps so that they, not the communicator set-up, precede the timed region
        gatherer.gather(torch.zeros(B, 1, device=dev))
        gatherer.synchronize()
        dist.barrier()
    with torch.no_grad():
        for i in range(max(args.warmup, nstreams if args.warmup else 0)):
            out = step(i)
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        t_sub = time.perf_counter()
        torch.cuda.synchronize()                 # every stream of the device: compute pipeline and gathers
        if os.environ.get("FVP_BENCH_DEBUG"):
            print(f"[debug] submit {1e3 * (t_sub - t0):.2f} ms, drain {1e3 * (time.perf_counter() - t_sub):.2f} ms", file=sys.stderr)
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # transparency: the strictly serial rate (one batch at a time on the current stream), 5 steps
        serial_fps = None
        if nstreams > 1 and graphed is None:
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for i in range(5):
                step(i, pipelined=False)
            torch.cuda.synchronize()
            serial_fps = 5 * B * world / (time.perf_counter() - ts)
        # per-class kernel timers (HIP events on the launch stream) in their own untimed steps:
        # the event pairs cost ~2 % and would serialise nothing but still perturb the timed region
        if not args.no_prof and graphed is None:
            lib.fvp_prof_reset()
            lib.fvp_prof_enable(2)                # one event pair per launch, every class
            for i in range(max(1, args.prof_steps)):
                step(i, pipelined=False)
            torch.cuda.synchronize()
    lib.fvp_prof_enable(0)
    prof_steps = max(1, args.prof_steps)
    if world > 1 or force_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape[0] == B * world
    valid_people = float((out[..., 0, 3] >= 0).sum().item()) / out.shape[0]

    if rank == 0:
        names = {capi.K_PROJECT_WHOLE: "project_whole", capi.K_PROJECT_TRIPLANE: "project_triplane",
                 capi.K_CONV: "conv_other", capi.K_CONV_WINO: "conv_winograd_3x3",
                 capi.K_SOFTARGMAX: "softargmax_weightnet", capi.K_OTHER: "other"}
        kern = {}
        if not args.no_prof:
            for cls, nm in names.items():
                ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
                lib.fvp_prof_read(cls, C.byref(ms), C.byref(n), C.byref(fl))
                kern[nm] = {"ms_total": ms.value, "launches": n.value, "flops": fl.value}
        J, V = cfg.DATASET.NUM_JOINTS, cfg.DATASET.CAMERA_NUM
        W, H = cfg.DATASET.HEATMAP_SIZE
        Cn = cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS[0]
        roof = None
        if kern:
            wino, conv, proj = kern["conv_winograd_3x3"], kern["conv_other"], kern["project_triplane"]

            def tf(k):
                return k["flops"] / (k["ms_total"] * 1e-3) / 1e12 if k["ms_total"] > 0 else 0.0

            pbytes = algorithmic_bytes_projection(V, J, H, W, Cn, valid_people) * B * prof_steps
            proj_gbs = pbytes / (proj["ms_total"] * 1e-3) / 1e9 if proj["ms_total"] > 0 else 0.0
            # headline roofline = the kernel with the largest accumulated time.  FLOPs are the
            # ALGORITHMIC ones (direct-conv 2*MAC); the Winograd kernel executes
            # 16/36 of them on the matrix cores.
            cands = {"k_conv_wino": wino["ms_total"], "k_conv_dma": conv["ms_total"],
                     "k_project_triplane": proj["ms_total"]}
            top = max(cands, key=cands.get)
            if top == "k_project_triplane":
                roof = {"kernel": "k_project_triplane", "bound": "hbm", "achieved": proj_gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": proj_gbs / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_us": 1e3 * proj["ms_total"] / max(proj["launches"], 1)}
            else:
                k = wino if top == "k_conv_wino" else conv
                roof = {"kernel": ("k_conv_wino (3x3 convs as Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32; P2PNet "
                                   "res-blocks)" if top == "k_conv_wino" else
                                   "k_conv_dma (fp32 MFMA implicit GEMM: 7x7, 1x1, transposed and small-map 3x3 convs)"),
                        "bound": "mfma", "achieved": tf(k), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": tf(k) / MFMA_F32_PEAK_TF, "traffic": None,
                        "avg_launch_us": 1e3 * k["ms_total"] / max(k["launches"], 1),
                        "flops": "algorithmic (direct conv 2*MAC); executed MFMA flops = 4/9 of that"
                                 if top == "k_conv_wino" else "algorithmic = executed"}
            # HBM-side traffic of the same kernel from the committed rocprofv3 PMC passes
            # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, mean per launch), if present
            # (newest profiles/rNN_pmc_traffic.json; PMC counters cannot be read from inside this process,
            # so the figure is tagged with the file, the commit it was collected at and the date)
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            if files:
                try:
                    with open(files[-1]) as f:
                        t = json.load(f)
                    key = {"k_conv_wino": "conv_wino_bytes_per_launch", "k_conv_dma": "conv_dma_bytes_per_launch",
                           "k_project_triplane": "project_triplane_bytes_per_launch"}[top]
                    roof["traffic"] = t.get(key)
                    roof["traffic_source"] = (f"profiles/{os.path.basename(files[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                              f"separate passes, tools/gpu_profile_all.sh; collected {t.get('collected', '?')} "
                                              f"at commit {t.get('commit', '?')}) - not measured in this run")
                except Exception:
                    pass
            wino["tflops_algorithmic"] = tf(wino)
            conv["tflops"] = tf(conv)
            allconv = {"ms_total": wino["ms_total"] + conv["ms_total"], "launches": wino["launches"] + conv["launches"],
                       "flops": wino["flops"] + conv["flops"]}
            allconv["tflops_algorithmic"] = tf(allconv)
            kern["conv_all"] = allconv
            kern["project_triplane"]["algorithmic_GBps"] = proj_gbs
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.config, 100, args.cpu_budget)
        mpjpe = mpjpe_vs_reference(dev) if not args.backbone else None
        frames = B * world * args.steps
        line = {
            "metric": "frames/sec at 5-view 80x80x20 voxel (heatmaps -> 3D joints); MPJPE vs ref (mm) in mpjpe_vs_ref_mm",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}-shape 5-view synthetic heatmaps, 80x80x20, jln64, "
                                   f"{B} frames/GPU/step, {valid_people:.1f} valid people/frame (MIN_SCORE=-1), "
                                   "seeded random weights", "frames_per_gpu_per_step": B,
                       "parallelism": f"frame-sharded dp{world}, all_gather of results on a dedicated stream behind each "
                                      f"batch's completion event",
                       "distinct_input_batches": nin,
                       "launch": "hipGraph replay" if args.graph else "eager (ctypes launches on the current stream)",
                       "batches_in_flight": nstreams,
                       "frames_per_s_one_batch_at_a_time": serial_fps,
                       "input": ("5 x [3,512,960] images per frame -> bf16 Pose-ResNet-50 -> voxel path" if args.backbone
                                 else "heatmaps resident in HBM")},
            "mpjpe_vs_ref_mm": mpjpe, "roofline": roof, "cpu_baseline": cpu, "kernels": kern,
        }
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: RCCL prints a version banner through C stdio (buffered until
        # exit when stdout is a pipe), so flush the C streams before printing
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
