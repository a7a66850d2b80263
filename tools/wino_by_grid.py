#!/usr/bin/env python
"""k_conv_wino launches of a rocprofv3 kernel trace split by launch size:

    python tools/wino_by_grid.py <trace_kernel_trace.csv> [n_workgroup_slots=256]

Since round 6 CenterNet's 80- / 40-wide levels run the same Winograd instances as P2PNet's res-blocks, but as launches of a
few workgroups (launch-latency-bound).  `rocprofv3 --stats` averages both under one kernel name; this prints, per template
instance, the chip-filling launches (>= n slots workgroups: the ones bench.py's roofline describes, class FVP_K_CONV_WINO) and
the sub-chip ones (FVP_K_CONV_WINO_SMALL) separately."""
import collections
import csv
import sys

path = sys.argv[1]
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 256
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "k_conv_wino" not in n:
        continue
    wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    # 4-wave workgroups run two per CU: their slot count is twice the CU count
    full = wgs >= slots * (2 if int(r["Workgroup_Size_X"]) == 256 else 1)
    acc[(n.split("(")[0].replace("void fvp::", ""), "chip-filling" if full else "sub-chip")].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.defaultdict(lambda: [0, 0.0])
for (name, kind), d in sorted(acc.items()):
    print(f"{name:44s} {kind:12s} calls {len(d):4d}  avg {sum(d) / len(d):8.1f} us  min {min(d):7.1f}  max {max(d):7.1f}")
    tot[kind][0] += len(d)
    tot[kind][1] += sum(d)
for kind, (n, t) in tot.items():
    print(f"ALL k_conv_wino, {kind:12s}: calls {n:4d}  avg {t / n:8.1f} us  total {t / 1e3:8.2f} ms")
