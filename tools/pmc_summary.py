#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_<tag>_*/p_counter_collection.csv): per kernel
name, mean counter value per dispatch and mean duration.  FETCH_SIZE / WRITE_SIZE are reported in
KiB by rocprofv3; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section) -- both raw and doubled figures are printed."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
SLOTS = 256                  # workgroup slots of the chip for 8-wave workgroups (one per CU)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, f"pmc_{tag}_*", "p_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("fvp::"):
            continue
        if "k_conv_wino" in name:
            # round 6: CenterNet's 80- / 40-wide levels run Winograd instances too, as launches of a few workgroups
            # (launch-latency-bound): kept apart from the chip-filling launches the roofline describes
            wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            if wgs < SLOTS * (2 if int(r["Workgroup_Size"]) == 256 else 1):
                name += " [sub-chip launch]"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc, key=lambda n: -sum(dur[n]))
for n in rows:
    d = sum(dur[n]) / len(dur[n])
    print(f"{n}  mean {d:.1f} us  x{len(dur[n]) // max(1, len(glob.glob(os.path.join(root, f'pmc_{tag}_*'))))} per pass-set")
    c = {k: sum(v) / len(v) for k, v in acc[n].items()}
    line = "   " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(c.items()))
    print(line)
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        f, w = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        print(f"   HBM-side traffic per launch: fetch {f / 1024:.2f} MiB (x2 correction: {2 * f / 1024:.2f} MiB), "
              f"write {w / 1024:.2f} MiB")

# machine-readable traffic summary for bench.py's roofline.traffic: one entry per kernel CLASS, averaged over the
# launches of every template variant seen (all of them listed, none silently dropped).  The profiled command runs the
# B = 8 step only (bench.py --no-mpjpe --no-extra): no fixture-sized launches enter the averages (round 2's file mixed
# them in, and kept only the last-seen variant for the projection kernels).
import datetime
import json

CLASSES = {"k_conv_wino": lambda n: "k_conv_wino" in n and "sub-chip" not in n,
           "k_conv_wino_sub_chip": lambda n: "k_conv_wino" in n and "sub-chip" in n,
           "k_conv_dma": lambda n: "k_conv" in n and "k_conv_wino" not in n and "k_conv1d" not in n,
           "k_conv1d_fused": lambda n: "k_conv1d" in n,
           "k_project_triplane": lambda n: "k_project_triplane" in n,
           "k_project_whole": lambda n: "k_project_whole" in n}
out = {"workload": {"config": os.environ.get("FVP_PMC_CONFIG", "panoptic"),
                    "frames_per_step": int(os.environ.get("FVP_PMC_BATCH", "8"))},
       "classes": {}}
for cls, match in CLASSES.items():
    variants, tot_b, tot_n = {}, 0.0, 0
    for n in rows:
        if not match(n):
            continue
        fs, ws = acc[n].get("FETCH_SIZE", []), acc[n].get("WRITE_SIZE", [])
        if not fs or not ws:
            continue
        f, w = sum(fs) / len(fs), sum(ws) / len(ws)
        b = (2 * f + w) * 1024
        variants[n] = {"bytes_per_launch": b, "fetch_KiB_raw": f, "write_KiB": w, "launches": len(fs),
                       "mean_us": sum(dur[n]) / len(dur[n])}
        tot_b += b * len(fs)
        tot_n += len(fs)
    if tot_n:
        out["classes"][cls] = {"bytes_per_launch": tot_b / tot_n, "launches": tot_n, "variants": variants}
out["note"] = ("bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts "
               "128-B requests as 64 B); per class = launch-weighted mean over the listed variants")
out["collected"] = datetime.date.today().isoformat()
out["commit"] = sys.argv[3] if len(sys.argv) > 3 else "?"
out["how"] = ("tools/gpu_pmc.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
              "`bench.py --steps 3 --streams 1 --no-mpjpe --no-extra --no-cpu-baseline --no-prof`")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print("wrote", sys.argv[2])
