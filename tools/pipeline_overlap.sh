#!/usr/bin/env bash
# Kernel-trace of the pipelined default bench (four batches in flight): how much of the wall time at least one kernel
# is running, how many run concurrently, and what the wall time is made of (each instant split equally between the kernels
# running in it, summed per kernel class).   gpurun -- 'bash tools/pipeline_overlap.sh [streams]'
streams="${1:-4}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/overlap"; rm -rf "$out"; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python "$root/bench.py" --steps 40 --warmup 5 --streams "$streams" --no-cpu-baseline --no-prof --no-mpjpe --no-extra > "$out/bench.log" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$streams" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
# steady-state window: from the first whole-space projection after the 5 warm-up steps to the one that opens the
# step after the 40 timed ones (one k_project_whole_q per step)
starts = [s for s, e, n in ev if "k_project_whole_q" in n]
t0, t1 = starts[5], starts[45]
sel = [(s, e) for s, e, n in ev if e > t0 and s < t1]
pts = []
for s, e in sel:
    pts.append((max(s, t0), 1)); pts.append((min(e, t1), -1))
pts.sort()
busy = 0; area = 0; cur = 0; last = t0; hist = {}
for t, d in pts:
    dt = t - last
    if cur > 0: busy += dt
    area += cur * dt
    hist[cur] = hist.get(cur, 0) + dt
    cur += d; last = t
wall = t1 - t0
ksum = sum(min(e, t1) - max(s, t0) for s, e in sel)
print(f"streams {sys.argv[2]}: window {wall/1e6:.2f} ms for 40 steps = {wall/40e3:.1f} us/step; sum of kernel durations {ksum/40e3:.1f} us/step")
print(f"  at least one kernel running {100*busy/wall:.1f} % of the time; mean concurrency while busy {area/busy:.2f}")
print("  time share by number of concurrent kernels: " + ", ".join(f"{k}: {100*v/wall:.1f} %" for k, v in sorted(hist.items())))
# wall-time attribution: every instant is split equally between the kernels running in it
import re
def cls(n):
    n = n.replace("void ", "").replace("fvp::", "")
    m = re.match(r"(k_\w+)", n)
    return m.group(1) if m else n.split("(")[0][:40]
named = [(max(s, t0), min(e, t1), cls(n)) for s, e, n in ev if e > t0 and s < t1]
pts2 = []
for i, (s, e, c) in enumerate(named):
    pts2.append((s, 0, i)); pts2.append((e, 1, i))
pts2.sort()
running = set(); last = t0; share = {}; solo = {}
for t, kind, i in pts2:
    dt = t - last
    if running and dt > 0:
        w = dt / len(running)
        for j in running:
            share[named[j][2]] = share.get(named[j][2], 0.0) + w
    last = t
    if kind == 0: running.add(i)
    else: running.discard(i)
for s, e, c in named:
    solo[c] = solo.get(c, 0) + (e - s)
print("  wall-time attribution per step (us; equal split between concurrent kernels) | sum of the class's kernel durations per step (us):")
for c, v in sorted(share.items(), key=lambda kv: -kv[1]):
    print(f"    {c:34s} {v/40e3:8.1f} | {solo[c]/40e3:8.1f}")
PY
