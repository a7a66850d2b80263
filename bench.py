#!/usr/bin/env python
"""Throughput of the heatmaps -> 3-D joints hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config NAME] [--backbone]

A step = one pass of the whole hot path (staging, HDN, JLN, fusion) over one batch of B
synthetic frames per GPU: Panoptic shape set (5 views, 15 joints, 240x128 heatmaps, 80x80x20
voxels, jln64, MAX_PEOPLE 10), Gaussian-blob heatmaps resident in HBM, seeded random weights,
MIN_SCORE = -1 so that all 10 proposals per frame are valid (P = 10, the FLOP count
BASELINE.md quotes).  Frames shard across ranks (one process per GPU, weak scaling); the only
collective is an all_gather of the [B,10,15,5] result.  Rank 0 prints ONE JSON line.

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher and starts
N ranks of itself (one per GPU, RCCL rendezvous on 127.0.0.1); under ``torch.distributed.run`` the ranks are
taken from the environment as before.  The reference has no launcher of its own (run/validate.py:13 imports
torch.utils.data.distributed and never uses it; README.md:96).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP runtime setting (not read by libfvp_hip.so): hardware queues per process.  With the default of 4, the compute streams
# of the batches in flight, the result-gather stream and the default stream share queues; four batches in flight then
# measured SLOWER than three (2 756 vs 2 938 frames/s), with >= 5 queues faster (3 016-3 020).  Must be set before the HIP
# runtime initialises; an explicit value in the environment wins.  8 is enough for ONE pipeline; this process builds five
# (default line + Shelf / 128x128x32 / Campus / end-to-end legs, ~20 streams, never destroyed), and the later legs lost 5-10 %
# on shared queues again (Shelf 2 512 vs 2 678): 24.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
_STREAM_POOL = {}


def stream_pool(dev, n):
    """ONE set of compute streams per device for every pipeline this process builds (round 5): the legs run one after the
    other, and streams are never destroyed - with a fresh set per leg the later legs ended up on shared hardware queues."""
    import torch
    pool = _STREAM_POOL.setdefault(str(dev), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # dense fp32 MFMA peak
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak (backbone leg)
BACKBONE_GFLOP_PER_VIEW = 108.5  # Pose-ResNet-50 at 512x960
CONFIGS = ("panoptic", "shelf", "campus", "panoptic128")
STUB = os.environ.get("FVP_BENCH_STUB") == "1"   # CPU / gloo stand-in step: tests the launcher and the timing protocol


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--config", default="panoptic", choices=CONFIGS,
                    help="shape set: panoptic (BASELINE configs[1], the headline), shelf (configs[2]), campus "
                         "(configs[0] shape), panoptic128 (configs[3]: 128x128x32, jln128; use --batch 1 per GPU)")
    ap.add_argument("--cpu-budget", type=float, default=100.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--inputs", type=int, default=4, help="distinct resident input batches rotated through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-mpjpe", action="store_true",
                    help="skip the float-parity replay (profiled runs: keeps fixture-sized launches out of the counters)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary legs of the default line (other configs, B = 1 latency, long run, end to end)")
    ap.add_argument("--long-steps", type=int, default=300, help="steps of the secondary long run (value_long)")
    ap.add_argument("--streams", type=int, default=4,
                    help="batches in flight: step i runs on HIP stream i %% S with its own scratch buffers, so the "
                         "detection stage of one batch overlaps the joint stage of the previous one")
    ap.add_argument("--prof-steps", type=int, default=5,
                    help="extra single-stream steps after the timed region with the per-class HIP-event timers on")
    ap.add_argument("--backbone", action="store_true",
                    help="end-to-end variant (BASELINE configs[4] shape): the step starts from 5 x [3,512,960] "
                         "images per frame and runs the bf16 Pose-ResNet-50 backbone in front of the voxel path")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured hipGraph (per-kernel event timing is then unavailable)")
    return ap.parse_args(argv)


# ---- launcher -------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n ranks of this script (one per GPU) and wait for them; rank 0 owns stdout (the JSON line), the
    other ranks' stdout goes to stderr.  Returns the exit code (first failing rank's, else 0)."""
    if not STUB:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this box has {have}; nothing was run "
                  f"(use --gpus {max(have, 1)} here, or launch on an {n}-GPU node)", file=sys.stderr)
            return 3
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending.values():
                    q.terminate()
        time.sleep(0.05)
    return rc


# ---- CPU baseline ---------------------------------------------------------------------------------------------
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


def cpu_baseline(cfg_name, seed, budget_s=100.0):
    """The CPU oracle (a port of the reference's PyTorch path, pinned against the reference's
    golden vectors) on the host cores of this box, same workload (P = 10 valid people per frame).
    Protocol of SURVEY.md section 8d: 2 warm-ups (sampling grid pre-built) then 10 iterations per leg, per-stage
    ms.  Legs, in this order: 8 threads at B = 1 and B = 8, then all physical cores at B = 1.  (All cores at
    B = 8 is NOT run: oneDNN oversubscribes these small ops - all-core runs measured 7x slower than 8 threads in
    rounds 1-2 - and 12 passes of 8 frames at ~0.3 frames/s would take five minutes.)  A leg that would overrun
    ``budget_s`` stops early, never below 3 iterations, and reports the count it did."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fvp_oracle as O
    import fvp_synthetic as S
    cfg = S.make_cfg(cfg_name, device="cpu", min_score=-1.0)
    cams, seq = S.load_cameras(cfg_name)
    rt = S.resize_transform(cfg)
    heat8 = S.heatmaps_blobs(cfg, cams, seq, 8, people=4, seed=seed)
    orc = O.Oracle(cfg, S.fill_state_dict(O.reference_state_dict_shapes(cfg), seed=7))
    phys = physical_cores()
    legs = [(8, 1, 10), (8, 8, 10)] + ([(phys, 1, 10)] if phys != 8 else [])
    runs = []
    t_start = time.perf_counter()
    hdn_k, jln_k = "project_whole+center_net+nms+c2c (hdn)", "project_individual+p2p+softargmax+fusion (jln)"
    for threads, B, iters in legs:
        torch.set_num_threads(threads)
        heat = heat8[:B]
        meta = {"seq": [seq] * B}
        stages = {hdn_k: 0.0, jln_k: 0.0}
        done = 0
        with torch.no_grad():
            for it in range(-2, iters):                 # 2 warm-ups (build the sampling grid)
                if it == 0:
                    t0 = time.perf_counter()
                    stages = {hdn_k: 0.0, jln_k: 0.0}
                ta = time.perf_counter()
                _, _, centers, _ = orc.hdn(heat, meta, cams, rt)
                tb = time.perf_counter()
                mask = centers[:, :, 3] >= 0
                orc.jln(meta, heat, centers, mask, cams, rt)
                tc = time.perf_counter()
                stages[hdn_k] += tb - ta
                stages[jln_k] += tc - tb
                if it >= 0:
                    done = it + 1
                    if done >= 3 and time.perf_counter() - t_start > budget_s:
                        break
        dt = time.perf_counter() - t0
        runs.append({"threads": threads, "batch": B, "iterations": done, "frames_per_s": B * done / dt,
                     "ms_per_frame_by_stage": {k: 1e3 * v / (B * done) for k, v in stages.items()}})
    best = max(runs, key=lambda r: r["frames_per_s"])
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
            "sample": f"torch-CPU oracle on {cpu_model()} ({phys} physical cores, {os.cpu_count()} logical): 8 threads at "
                      f"B = 1 and B = 8, {phys} threads at B = 1; 10 iterations each after 2 warm-ups (fewer only if the "
                      f"{budget_s:.0f}-s budget ran out: see iterations), P = 10 people/frame; value = best of the runs below",
            "runs": runs}


def mpjpe_vs_reference(dev):
    """BASELINE.json's second metric half: the committed float-parity fixtures (outputs of the REFERENCE
    itself, tests/golden/make_golden.py) replayed through the HIP path; mean / max joint distance in mm."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import make_inputs, make_weights
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    out = {}
    for case in ("panoptic_c_b2_thr", "shelf_c_b1_thr", "campus_c_b2_thr"):
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
        if case.startswith("campus"):
            # Campus: the reference's own fp32 result is 1e-3 .. 2e-3 mm from its float64 evaluation on every person, so
            # the figure that says something is the build's distance to the float64 evaluation relative to the reference's
            f3 = fused[..., :3].cpu().numpy().astype(np.float64)
            d64 = np.linalg.norm(f3 - g["floor_fused"], axis=-1)
            pfl = np.linalg.norm(g["fused_poses"][..., :3].astype(np.float64) - g["floor_fused"], axis=-1).max(axis=-1)
            out[case].update(max_vs_fp64=float(d64[v].max()),
                             worst_build_vs_fp64_over_reference_vs_fp64=float((d64.max(axis=-1)[v] / pfl[v]).max()),
                             worst_build_vs_ref32_over_reference_vs_fp64=float(
                                 (np.linalg.norm(f3 - g["fused_poses"][..., :3], axis=-1).max(axis=-1)[v] / pfl[v]).max()),
                             bar="tests/common.py FLOOR_RULE: ratios <= 1.5 / 2.0 (1e-3 mm is below the reference's own "
                                 "fp32 reproducibility on this shape)")
    try:
        import seed_sweep                                   # tests/golden: >= 10 consecutive seeds per shape
        out.update(seed_sweep.replay_all(dev))
    except Exception as e:                                   # fixtures absent: the two cases above still stand
        out["seed_sweep"] = {"error": repr(e)}
    first = out["panoptic_c_b2_thr"]
    # one line per shape: the conditioned fixture where there is one, the well-conditioned proposals of the seed sweep
    # (rule R1p of tests/golden/seed_sweep.py) for jln128
    def sw(name):
        e = out.get("sweep_" + name) or {}
        return {"joints": e.get("joints_of_proposals_with_floor_le_4e-4"), "max": e.get("max_mm_in_proposals_with_floor_le_4e-4"),
                "frac_of_all_joints_within_1e-3": e.get("frac_within_1e-3_mm")}
    by_shape = {"panoptic (jln64)": {"max": first["max"], "mean": first["mean"], "bar": 1e-3},
                "shelf": {"max": out["shelf_c_b1_thr"]["max"], "mean": out["shelf_c_b1_thr"]["mean"], "bar": 1e-3},
                "campus": {"max": out["campus_c_b2_thr"]["max"], "reference_own_fp32_vs_fp64_max": out["campus_c_b2_thr"]["reference_fp32_vs_fp64_floor_max"],
                           "build_vs_fp64_over_reference_vs_fp64": out["campus_c_b2_thr"].get("worst_build_vs_fp64_over_reference_vs_fp64"),
                           "bar": "ratio <= 1.5 (FLOOR_RULE)"},
                "panoptic128 (jln128, 10-seed sweep, well-conditioned proposals)": dict(sw("panoptic128_b1"), bar=1e-3),
                "panoptic B = 8 (10-seed sweep, well-conditioned proposals)": dict(sw("panoptic_b8"), bar=1e-3)}
    return {"fixture": "panoptic_c_b2_thr (Panoptic 5-view 80x80x20 jln64, 8 valid people; reference outputs committed "
                       "under tests/golden)", "mean": first["mean"], "max": first["max"], "bar": 1e-3,
            "by_shape": by_shape, "all": out}


# ---- the workload -----------------------------------------------------------------------------------------------
class Workload:
    """Model + resident inputs + pipeline of one (config, batch, depth) on this rank's GPU."""

    def __init__(self, config, B, dev, rank, nin, nstreams, backbone, graph, gatherer):
        import torch
        import fvp_synthetic as S
        from faster_voxelpose_amd.models import faster_voxelpose as FV
        self.torch, self.B, self.dev, self.gatherer = torch, B, dev, gatherer
        cfg = self.cfg = S.make_cfg(config, device=dev, min_score=-1.0)
        self.cams, seq = S.load_cameras(config)
        self.rt = S.resize_transform(cfg).to(dev)
        # weak scaling: every rank owns B frames per step (rank r = frames [r*B, (r+1)*B) of the global batch);
        # `nin` distinct batches per rank stay resident in HBM and are rotated through the steps
        self.nin = nin = max(1, nin)
        self.heats = [S.heatmaps_blobs(cfg, self.cams, seq, B, people=4, seed=100 + 16 * rank + i).to(dev)
                      for i in range(nin)]
        self.meta = {"seq": [seq] * B}
        self.model = FV.get(cfg).to(dev)
        self.model.load_state_dict(S.fill_state_dict(self.model.state_dict(), seed=7))
        self.nstreams = max(1, nstreams)
        # batches in flight: FV.PipelinedForward = one replica (scratch buffers) + one HIP stream each
        self.pipe = (FV.PipelinedForward(self.model, depth=self.nstreams, streams=stream_pool(dev, self.nstreams))
                     if self.nstreams > 1 else None)
        self.bb = self.views_all = self.graphed = None
        if backbone:
            from faster_voxelpose_amd.core import config as CFG
            from faster_voxelpose_amd.models import resnet as RN
            self.bb = RN.get(CFG.default_config()).to(dev)
            self.bb.load_state_dict(S.fill_backbone_state_dict(self.bb.state_dict(), seed=3))
            Wi, Hi = cfg.DATASET.IMAGE_SIZE
            self.views_all = [torch.rand(B, cfg.DATASET.CAMERA_NUM, 3, Hi, Wi, device=dev) for _ in range(min(nin, 2))]
        if graph:
            self.graphed = FV.GraphedForward(self.model, self.meta, self.heats[0], self.cams, self.rt)

    def step(self, i=0, pipelined=True):
        torch = self.torch
        if self.bb is not None:
            kw = dict(backbone=self.bb, views=self.views_all[i % len(self.views_all)])
        else:
            kw = dict(input_heatmaps=self.heats[i % self.nin])
        if self.graphed is not None:
            # the graph's static output is overwritten by the next replay: hand the gather a private copy,
            # ordered behind this replay on the current stream (ResultGatherer waits for that stream)
            return self.gatherer.gather(self.graphed(self.heats[i % self.nin])[0].clone())
        if self.pipe is not None and pipelined:
            (fused, _, _, _, _), ev = self.pipe.submit(meta=self.meta, cameras=self.cams, resize_transform=self.rt, **kw)
            return self.gatherer.gather(fused, ev)
        fused = self.model(meta=self.meta, cameras=self.cams, resize_transform=self.rt, **kw)[0]
        ev = torch.cuda.Event()
        ev.record()
        return self.gatherer.gather(fused, ev)

    def sync(self):
        self.gatherer.synchronize()              # host-issued gathers still waiting for their batch are enqueued here
        self.torch.cuda.synchronize()


class StubWorkload:
    """FVP_BENCH_STUB=1: a per-frame-independent CPU function in place of the hot path (gloo ranks), so the
    launcher, the barrier / max-over-ranks timing and the JSON contract can be tested without a GPU."""

    def __init__(self, B, rank, gatherer):
        import torch
        self.torch, self.B, self.rank, self.gatherer = torch, B, rank, gatherer
        self.nstreams = 1

    def step(self, i=0, pipelined=True):
        torch = self.torch
        frames = torch.arange(self.rank * self.B, (self.rank + 1) * self.B, dtype=torch.float32)
        fused = (frames.view(-1, 1, 1, 1) + torch.zeros(1, 10, 15, 5)).contiguous()
        fused[..., 3] = 0.0
        return self.gatherer.gather(fused)

    def sync(self):
        pass


HOST_TIMES = {}       # of the most recent timed_region: host submit ms per step, drain ms


def timed_region(wl, steps, warmup, dist_on, dev):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device synchronisation on both
    sides; returns this rank's seconds and the last step's (gathered) output."""
    import torch
    import torch.distributed as dist
    out = None
    with torch.no_grad():
        for i in range(max(warmup, wl.nstreams if warmup else 0)):
            out = wl.step(i)
        wl.sync()
        if dist_on:
            dist.barrier()
        wl.sync()
        pipe = getattr(wl, "pipe", None)
        w0 = pipe.wait_s if pipe is not None else 0.0
        t0 = time.perf_counter()
        for i in range(steps):
            out = wl.step(i)
        t_sub = time.perf_counter()
        waited = (pipe.wait_s - w0) if pipe is not None else 0.0
        wl.sync()                                # every stream of the device: compute pipeline and gathers
        # host time spent ENQUEUEING the K steps (the pipeline's back-pressure wait - the host blocked until the slot's
        # previous batch had finished - is reported apart), and how long the GPU still ran after the last enqueue
        HOST_TIMES.update(submit_ms_per_step=1e3 * (t_sub - t0 - waited) / max(1, steps),
                          backpressure_wait_ms_per_step=1e3 * waited / max(1, steps),
                          drain_ms=1e3 * (time.perf_counter() - t_sub))
        if os.environ.get("FVP_BENCH_DEBUG"):
            print(f"[debug] submit {1e3 * (t_sub - t0):.2f} ms, drain {1e3 * (time.perf_counter() - t_sub):.2f} ms",
                  file=sys.stderr)
        if dist_on:
            dist.barrier()
        wl.sync()
        dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def serial_rate(wl, world, steps=5):
    import torch
    with torch.no_grad():
        wl.step(0, pipelined=False)
        wl.sync()
        ts = time.perf_counter()
        for i in range(steps):
            wl.step(i, pipelined=False)
        wl.sync()
    return steps * wl.B * world / (time.perf_counter() - ts)


def secondary_leg(config, B, streams, steps, warmup, dev, backbone=False, serial=True):
    """One more (config, batch) through the same protocol on this GPU; world size 1 only."""
    import torch
    from faster_voxelpose_amd.core import distributed as D
    wl = Workload(config, B, dev, 0, 4, streams, backbone, False, D.ResultGatherer(1, device=dev))
    dt, out = timed_region(wl, steps, warmup, False, dev)
    leg = {"config": config, "frames_per_step": B, "batches_in_flight": wl.nstreams, "steps": steps,
           "frames_per_s": B * steps / dt, "ms_per_step": 1e3 * dt / steps,
           "valid_people_per_frame": float((out[..., 0, 3] >= 0).sum().item()) / out.shape[0],
           "host_submit_ms_per_step": HOST_TIMES.get("submit_ms_per_step"), "drain_ms": HOST_TIMES.get("drain_ms")}
    if serial:
        leg["frames_per_s_one_batch_at_a_time"] = serial_rate(wl, 1)
    del wl
    torch.cuda.empty_cache()
    return leg


def graph_slots_leg(config, B, depth, steps, dev):
    """The same pipeline with every slot captured as a hipGraph (FV.GraphedPipeline): frames/s and host submit time."""
    import torch
    import fvp_synthetic as S
    from faster_voxelpose_amd.models import faster_voxelpose as FV
    cfg = S.make_cfg(config, device=dev, min_score=-1.0)
    cams, seq = S.load_cameras(config)
    rt = S.resize_transform(cfg).to(dev)
    heats = [S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100 + i).to(dev) for i in range(4)]
    model = FV.get(cfg).to(dev)
    model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
    with torch.no_grad():
        gp = FV.GraphedPipeline(model, depth, {"seq": [seq] * B}, heats[0], cams, rt, streams=stream_pool(dev, depth))
        for i in range(2 * depth):
            gp.submit(heats[i % 4])
        gp.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            gp.submit(heats[i % 4])
        t1 = time.perf_counter()
        gp.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    del gp, model
    torch.cuda.empty_cache()
    return {"config": config, "frames_per_step": B, "batches_in_flight": depth, "steps": steps, "launch": "one hipGraph per pipeline slot",
            "frames_per_s": B * steps / (t2 - t0), "ms_per_step": 1e3 * (t2 - t0) / steps,
            "host_submit_ms_per_step": 1e3 * (t1 - t0) / steps, "drain_ms": 1e3 * (t2 - t1)}


def load_pmc_traffic(top, B, config):
    """HBM-side bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary, or
    (None, reason).  Refused unless the file says it was collected on THIS workload (config, frames per step)
    and carries an entry for exactly this kernel class with the kernel variants it was averaged over."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.json"
    f = files[-1]
    try:
        with open(f) as fh:
            t = json.load(fh)
    except Exception as e:
        return None, f"{os.path.basename(f)} unreadable: {e!r}"
    wk = t.get("workload") or {}
    if wk.get("config") != config or wk.get("frames_per_step") != B:
        return None, (f"{os.path.basename(f)} was collected on workload {wk or 'unknown (pre-round-3 file)'}, this run is "
                      f"{config} B = {B}: refused")
    ent = (t.get("classes") or {}).get(top)
    if not ent or not ent.get("variants"):
        return None, f"{os.path.basename(f)} has no per-variant entry for {top}: refused"
    src = (f"profiles/{os.path.basename(f)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over "
           f"`bench.py --no-mpjpe --no-extra --streams 1`, tools/gpu_profile_all.sh; kernel variants "
           f"{sorted(ent['variants'])}; collected {t.get('collected', '?')} at commit {t.get('commit', '?')}) - "
           f"not measured in this run")
    return ent["bytes_per_launch"], src


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: pass --gpus equal to the number of ranks "
              f"(or drop WORLD_SIZE and let bench.py start the ranks itself)", file=sys.stderr)
        sys.exit(2)
    if STUB:
        dev = "cpu"
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} needs GPU {local_rank}; visible GPUs: "
                  f"{torch.cuda.device_count() if torch.cuda.is_available() else 0}", file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(local_rank)
        dev = f"cuda:{local_rank}"
    # FVP_BENCH_FORCE_DIST=1: initialise RCCL and run the result gather for a single rank too (a one-GPU check of
    # the N > 1 code path: process group, communication stream, all_gather_into_tensor)
    force_dist = world == 1 and os.environ.get("FVP_BENCH_FORCE_DIST") == "1"
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo" if STUB else "nccl", rank=rank, world_size=world)

    from faster_voxelpose_amd.core import distributed as D
    B = args.batch
    lo, hi = D.shard_frames(B * world, world, rank)
    assert hi - lo == B
    # the gather of batch t runs on its own stream behind batch t's completion event, so it never fences
    # the compute pipeline (core/distributed.py)
    # FVP_GATHER_ISSUE=stream (diagnostics): the pre-round-6 GPU-side event wait instead of the host-issued gather
    gatherer = D.ResultGatherer(world, device=dev, always=force_dist, issue=os.environ.get("FVP_GATHER_ISSUE", "host"))
    lib = capi = None
    if STUB:
        wl = StubWorkload(B, rank, gatherer)
        args.no_prof = args.no_cpu_baseline = args.no_mpjpe = args.no_extra = True
    else:
        from faster_voxelpose_amd import _capi as capi
        lib = capi.load()
        wl = Workload(args.config, B, dev, rank, args.inputs, args.streams, args.backbone, args.graph, gatherer)
        if args.backbone or args.graph:
            args.no_prof = True
    nstreams = wl.nstreams

    if dist_on:
        # RCCL builds its communicator lazily inside the first collective (seconds of host time with an idle GPU);
        # do that before the warm-up steps so that they, not the communicator set-up, precede the timed region
        gatherer.gather(torch.zeros(B, 1, device=dev))
        gatherer.synchronize()
        dist.barrier()
    dt, out = timed_region(wl, args.steps, args.warmup, dist_on, dev)
    headline_host = {"submit_ms_per_step": HOST_TIMES.get("submit_ms_per_step"), "drain_ms": HOST_TIMES.get("drain_ms"),
                     "backpressure_wait_ms_per_step": HOST_TIMES.get("backpressure_wait_ms_per_step"),
                     "what": "host time to enqueue one step of the timed region (Python + ctypes launches); time the host was "
                             "blocked by the pipeline's back-pressure (it may be at most `batches_in_flight` batches ahead of "
                             "the GPU: a long wait = the GPU, not the host, is the limit); and how long the GPU still ran "
                             "after the last enqueue"}
    assert out.shape[0] == B * world
    valid_people = float((out[..., 0, 3] >= 0).sum().item()) / out.shape[0]

    # transparency: the strictly serial rate (one batch at a time on the current stream), 5 steps
    serial_fps = None
    if not STUB and nstreams > 1 and not args.graph:
        serial_fps = serial_rate(wl, world)
    # secondary long run of the same step (>= 200 steady-state iterations)
    value_long = None
    if not STUB and not args.no_extra and world == 1 and args.long_steps > args.steps:
        dtl, _ = timed_region(wl, args.long_steps, 0, False, dev)
        value_long = {"frames_per_s": B * args.long_steps / dtl, "steps": args.long_steps,
                      "ms_per_step": 1e3 * dtl / args.long_steps}
    # per-class kernel timers (HIP events on the launch stream) in their own untimed steps:
    # the event pairs cost ~2 % and would serialise nothing but still perturb the timed region
    prof_steps = max(1, args.prof_steps)
    if not args.no_prof:
        with torch.no_grad():
            lib.fvp_prof_reset()
            lib.fvp_prof_enable(2)                # one event pair per launch, every class
            for i in range(prof_steps):
                wl.step(i, pipelined=False)
            torch.cuda.synchronize()
    if lib is not None:
        lib.fvp_prof_enable(0)

    line = None
    if rank == 0:
        kern, roof = {}, None
        if not args.no_prof:
            names = {capi.K_PROJECT_WHOLE: "project_whole", capi.K_PROJECT_TRIPLANE: "project_triplane",
                     capi.K_CONV: "conv_other", capi.K_CONV_WINO: "conv_winograd_3x3",
                     # Winograd launches with fewer work units than workgroup slots (CenterNet's 80- / 40-wide levels since
                     # round 6): launch-latency-bound, kept out of the dominant kernel's roofline
                     capi.K_CONV_WINO_SMALL: "conv_winograd_3x3_sub_chip_launches",
                     capi.K_SOFTARGMAX: "softargmax_weightnet", capi.K_OTHER: "other"}
            for cls, nm in names.items():
                ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
                lib.fvp_prof_read(cls, C.byref(ms), C.byref(n), C.byref(fl))
                kern[nm] = {"ms_total": ms.value, "launches": n.value, "flops": fl.value}
            cfg = wl.cfg
            J, V = cfg.DATASET.NUM_JOINTS, cfg.DATASET.CAMERA_NUM
            W, H = cfg.DATASET.HEATMAP_SIZE
            Cn = cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS[0]
            wino, conv, proj = kern["conv_winograd_3x3"], kern["conv_other"], kern["project_triplane"]

            def tf(k):
                return k["flops"] / (k["ms_total"] * 1e-3) / 1e12 if k["ms_total"] > 0 else 0.0

            psec = proj["ms_total"] * 1e-3
            pbytes = algorithmic_bytes_projection(V, J, H, W, Cn, valid_people) * B * prof_steps
            proj_gbs = pbytes / psec / 1e9 if psec > 0 else 0.0
            # the fused projection is latency / L2-bound, so its gather rate is reported too:
            # P*V*J*C^3 bilinear samples per person (every voxel of the person's cube, every view, every joint);
            # a (voxel, view) pair is one 4-tap fetch of JP contiguous channels
            samples = valid_people * B * prof_steps * float(V) * J * Cn ** 3
            proj["algorithmic_GBps"] = proj_gbs
            proj["samples_per_s"] = samples / psec if psec > 0 else 0.0
            proj["voxel_view_taps_per_s"] = samples / J / psec if psec > 0 else 0.0
            # headline roofline = the kernel with the largest accumulated time.  FLOPs are the
            # ALGORITHMIC ones (direct-conv 2*MAC); the Winograd kernel executes
            # 16/36 of them on the matrix cores.
            cands = {"k_conv_wino": wino["ms_total"], "k_conv_dma": conv["ms_total"],
                     "k_project_triplane": proj["ms_total"]}
            top = max(cands, key=cands.get)
            if top == "k_project_triplane":
                roof = {"kernel": "k_project_triplane", "bound": "hbm", "achieved": proj_gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": proj_gbs / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_us": 1e3 * proj["ms_total"] / max(proj["launches"], 1),
                        "samples_per_s": proj["samples_per_s"]}
            else:
                k = wino if top == "k_conv_wino" else conv
                # the kernel's matrix-pipe utilisation: EXECUTED MFMA FLOPs / time / fp32 MFMA peak.  Winograd F(2x2,3x3)
                # executes 16 multiplies per 2x2 outputs instead of 36, i.e. 4/9 of the algorithmic (direct-conv 2*MAC,
                # SURVEY.md 8d) FLOPs; the algorithmic rate - work delivered per second - is reported beside it and may
                # exceed the peak, which is the point of the algorithm, not a utilisation
                ex = 4.0 / 9.0 if top == "k_conv_wino" else 1.0
                roof = {"kernel": ("k_conv_wino (3x3 convs as Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32; P2PNet "
                                   "res-blocks)" if top == "k_conv_wino" else
                                   "k_conv_dma (fp32 MFMA implicit GEMM: 7x7, 1x1, transposed and small-map 3x3 convs)"),
                        "bound": "mfma", "achieved": tf(k) * ex, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": tf(k) * ex / MFMA_F32_PEAK_TF, "traffic": None,
                        "avg_launch_us": 1e3 * k["ms_total"] / max(k["launches"], 1),
                        "flops": ("EXECUTED on the matrix cores = 4/9 of the algorithmic direct-conv 2*MAC count (Winograd)"
                                  if top == "k_conv_wino" else "algorithmic = executed"),
                        "achieved_algorithmic": tf(k), "frac_algorithmic": tf(k) / MFMA_F32_PEAK_TF}
            # HBM-side traffic of the same kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
            # correction + WRITE_SIZE, mean per launch over the B = 8 launches of this very workload).  PMC counters
            # cannot be read from inside this process, so the figure is tagged with file, commit and date.
            roof["traffic"], roof["traffic_source"] = load_pmc_traffic(top, B, args.config)
            wino["tflops_algorithmic"] = tf(wino)
            conv["tflops"] = tf(conv)
            wsm = kern["conv_winograd_3x3_sub_chip_launches"]
            allconv = {"ms_total": wino["ms_total"] + conv["ms_total"] + wsm["ms_total"],
                       "launches": wino["launches"] + conv["launches"] + wsm["launches"],
                       "flops": wino["flops"] + conv["flops"] + wsm["flops"]}
            allconv["tflops_algorithmic"] = tf(allconv)
            kern["conv_all"] = allconv
            kern["per_step_ms"] = {nm: kern[nm]["ms_total"] / prof_steps for nm in names.values()}

        # ---- secondary legs (world size 1, default run): the other BASELINE configs through the same protocol
        other, latency_b1, e2e, pipe_legs = None, None, None, None
        if not args.no_extra and world == 1 and not STUB and not args.backbone and not args.graph:
            del wl
            torch.cuda.empty_cache()
            other = {}
            try:
                l1 = secondary_leg(args.config, 1, 1, 30, 5, dev, serial=False)
                latency_b1 = {"ms": l1["ms_per_step"], "frames_per_s": l1["frames_per_s"],
                              "what": "B = 1, one batch at a time on one stream (strictly serial), 30 steps"}
                if args.config != "shelf":
                    other["shelf (BASELINE configs[2]): 5 views, J = 17, 80x80x20, jln64, B = 8"] = \
                        secondary_leg("shelf", 8, args.streams, 20, 3, dev)
                if args.config != "panoptic128":
                    other["panoptic128 (BASELINE configs[3]): 128x128x32, jln128, one frame per GPU (B = 1)"] = \
                        secondary_leg("panoptic128", 1, args.streams, 20, 3, dev)
                if B != 32:
                    other[f"{args.config}, B = 32 per step (same shape, larger batch: parity = the sweep_panoptic_b32 leg)"] = \
                        secondary_leg(args.config, 32, args.streams, 10, 2, dev)
                if args.config != "campus":
                    other["campus (BASELINE configs[0] shape): 3 views, J = 17, 80x80x20, jln64, B = 8"] = \
                        secondary_leg("campus", 8, args.streams, 20, 3, dev)
                # launch path: B = 1 with four batches in flight (the reference's demo batch) and the headline shape, eager
                # launches against one hipGraph per slot - is the host the limit?
                pipe_legs = {"b1_eager": secondary_leg(args.config, 1, args.streams, 100, 8, dev, serial=False),
                             "b1_graph_slots": graph_slots_leg(args.config, 1, args.streams, 100, dev),
                             f"b{B}_graph_slots": graph_slots_leg(args.config, B, args.streams, 40, dev)}
                leg = secondary_leg("panoptic", 8, 3, 12, 3, dev, backbone=True)
                gf = BACKBONE_GFLOP_PER_VIEW * 5
                e2e = dict(leg, what="BASELINE configs[4] shape on one GPU: 5 x [3,512,960] images per frame -> bf16 "
                                     "Pose-ResNet-50 (v_mfma_f32_32x32x16_bf16; stem + max-pool and layer1's bottlenecks as fused kernels) -> voxel path, "
                                     "B = 8, 3 batches in flight",
                           backbone_gflop_per_frame=gf)
                # backbone alone (serial, HIP-event timed by the library) for its own roofline fraction
                e2e.update(backbone_alone(dev))
            except Exception as e:                              # a failing secondary leg must not cost the headline
                other["error"] = repr(e)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.config, 100, args.cpu_budget)
        mpjpe = mpjpe_vs_reference(dev) if not (args.backbone or args.no_mpjpe) else None
        frames = B * world * args.steps
        nine = "80x80x20, jln64" if args.config != "panoptic128" else "128x128x32, jln128"
        line = {
            "metric": "frames/sec at 5-view 80x80x20 voxel (heatmaps -> 3D joints); MPJPE vs ref (mm) in mpjpe_vs_ref_mm",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not args.backbone else "bf16 backbone + f32 voxel path", "data": "synthetic",
            "config": {"workload": ("stub step (FVP_BENCH_STUB=1): launcher / timing protocol check, no GPU" if STUB else
                                    f"{args.config}-shape {5 if args.config != 'campus' else 3}-view synthetic heatmaps, "
                                    f"{nine}, {B} frames/GPU/step, {valid_people:.1f} valid people/frame (MIN_SCORE=-1), "
                                    "seeded random weights"), "frames_per_gpu_per_step": B,
                       "parallelism": f"frame-sharded dp{world}, all_gather of results on a dedicated stream behind each "
                                      f"batch's completion event",
                       "distinct_input_batches": args.inputs,
                       "launch": "hipGraph replay" if args.graph else "eager (ctypes launches on the current stream)",
                       "batches_in_flight": nstreams,
                       "frames_per_s_one_batch_at_a_time": serial_fps,
                       # the same step over --long-steps (300) steps: the driver's timed region of 20 steps is ~50 ms and
                       # box-to-box noise is +-2 %; kept inside `config` so that it survives parsers that keep the contract
                       # keys only (also at top level as value_long)
                       "long_run": value_long,
                       "input": ("5 x [3,512,960] images per frame -> bf16 Pose-ResNet-50 -> voxel path" if args.backbone
                                 else "heatmaps resident in HBM")},
            "value_long": value_long, "latency_ms_b1_serial": latency_b1,
            "host": headline_host, "pipeline_launch_paths": pipe_legs,
            "mpjpe_vs_ref_mm": mpjpe, "roofline": roof, "cpu_baseline": cpu, "kernels": kern,
            "other_configs": other, "e2e": e2e,
        }
    if dist_on:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: RCCL prints a version banner through C stdio (buffered until
        # exit when stdout is a pipe), so flush the C streams before printing
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


def backbone_alone(dev, images=40, iters=3):
    """The bf16 Pose-ResNet-50 on 40 images (8 frames x 5 views), timed with HIP events around whole passes."""
    import torch
    import fvp_synthetic as S
    from faster_voxelpose_amd.core import config as CFG
    from faster_voxelpose_amd.models import resnet as RN
    bb = RN.get(CFG.default_config()).to(dev)
    bb.load_state_dict(S.fill_backbone_state_dict(bb.state_dict(), seed=3))
    x = torch.rand(images, 3, 512, 960, device=dev)
    with torch.no_grad():
        for _ in range(2):
            bb._run(x, True, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            bb._run(x, True, True)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tfs = BACKBONE_GFLOP_PER_VIEW * images / ms                 # GFLOP / ms = TFLOP/s
    del bb, x
    torch.cuda.empty_cache()
    return {"backbone_ms_per_40_images": ms, "backbone_tflops": tfs, "backbone_frac_of_bf16_peak": tfs / MFMA_BF16_PEAK_TF}


if __name__ == "__main__":
    main()
