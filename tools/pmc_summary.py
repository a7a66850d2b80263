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
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, f"pmc_{tag}_*", "p_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("fvp::"):
            continue
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

# machine-readable traffic summary for bench.py's roofline.traffic
import json
out = {}
grp = {"conv_wino": [0.0, 0.0, 0.0], "conv_dma": [0.0, 0.0, 0.0]}
for n in rows:
    c = {k: sum(v) / len(v) for k, v in acc[n].items()}
    nl = len(acc[n].get("FETCH_SIZE", []))
    g = "conv_wino" if "k_conv_wino" in n else ("conv_dma" if "k_conv" in n else None)
    if g and nl:
        grp[g][0] += c.get("FETCH_SIZE", 0.0) * nl
        grp[g][1] += c.get("WRITE_SIZE", 0.0) * len(acc[n].get("WRITE_SIZE", []))
        grp[g][2] += nl
    if "k_project_triplane" in n and nl:
        out["project_triplane_bytes_per_launch"] = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024
    if "k_project_whole" in n and nl:
        out["project_whole_bytes_per_launch"] = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024
for g, (f, w, n) in grp.items():
    if n:
        out[g + "_bytes_per_launch"] = (2 * f + w) * 1024 / n
        out[g + "_launches_sampled"] = n
out["note"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"
import datetime
out["collected"] = datetime.date.today().isoformat()
out["commit"] = sys.argv[3] if len(sys.argv) > 3 else "?"
out["how"] = "tools/gpu_profile_all.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 3 --streams 1"
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print("wrote", sys.argv[2])
