#!/usr/bin/env bash
# Kernel-trace of the pipelined default bench (three batches in flight): how much of the wall time at least one kernel
# is running, and how many run concurrently.   gpurun -- 'bash tools/pipeline_overlap.sh [streams]'
streams="${1:-3}"
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
PY
