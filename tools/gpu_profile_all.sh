#!/usr/bin/env bash
# Regenerates every profiles/rNN_* file in ONE GPU-box visit (then copy gpurun_out/profiles_<tag>/* to profiles/):
#   gpurun --timeout 1500 -- 'bash tools/gpu_profile_all.sh r02 <commit>'
# 1. GPU parity suite (parity report)      2. bench: default line + batch / pipeline-depth sweep + end-to-end
# 3. rocprofv3 --kernel-trace --stats of the serial B=8 step and of the backbone
# 4. PMC passes (SQ, GRBM, TCC, FETCH_SIZE, WRITE_SIZE: one group per pass) + summaries
tag="${1:-r06}"; commit="${2:-?}"
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; dst="$out/profiles_${tag}"
mkdir -p "$dst"
cd "$root"
export TMPDIR=/tmp
echo "== pytest -m gpu"; rm -f "$out/parity_report.jsonl"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$out/pytest_gpu_${tag}.log" 2>&1; echo "pytest rc=$?"; tail -3 "$out/pytest_gpu_${tag}.log"
cp "$out/parity_report.jsonl" "$dst/${tag}_parity_report.jsonl" 2>/dev/null
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 > "$dst/${tag}_bench_b8.json" 2> "$out/bench_${tag}.err"; echo "bench rc=$?"
for spec in "1 1" "1 4" "2 4" "8 1" "8 2" "8 3" "16 4" "32 4"; do
  set -- $spec
  timeout 300 python bench.py --steps 10 --warmup 3 --batch $1 --streams $2 --no-cpu-baseline --no-extra > "$dst/${tag}_bench_b$1_s$2.json" 2>> "$out/bench_${tag}.err"
done
timeout 300 python tools/bench_backbone.py --images 40 --iters 3 --per-op > "$dst/${tag}_backbone_per_op.log" 2>&1
for st in 1 2 3; do
  timeout 300 python bench.py --backbone --steps 8 --warmup 2 --streams $st --no-cpu-baseline --no-extra > "$dst/${tag}_bench_e2e_b8_s$st.json" 2>> "$out/bench_${tag}.err"
done
timeout 300 python tools/bench_conv.py --net conv_net --frames 8 --iters 10 > "$dst/${tag}_conv_per_op_p2pnet_b8.log" 2>&1
timeout 300 python tools/bench_conv.py --net center_net --frames 8 --iters 10 > "$dst/${tag}_conv_per_op_centernet_b8.log" 2>&1
python - "$dst" "$tag" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], sys.argv[2] + "_bench_*.json"))):
    try:
        d = json.load(open(f)); c = d["config"]
        print(os.path.basename(f), "frames/s %.1f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "serial", c.get("frames_per_s_one_batch_at_a_time"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}" -o trace -- python "$root/bench.py" --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-prof --no-mpjpe --no-extra > "$out/rocprof_${tag}.log" 2>&1; echo "rocprof rc=$?"
find "$out/prof_${tag}" -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} "$dst/${tag}_kernel_stats_b8.csv"
find "$out/prof_${tag}" -name "*kernel_trace.csv" | head -1 | xargs -r -I{} python "$root/tools/wino_by_grid.py" {} > "$dst/${tag}_kernel_stats_b8_wino_by_launch_size.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_bb" -o trace -- python "$root/tools/bench_backbone.py" --images 40 --iters 3 > "$out/rocprof_${tag}_bb.log" 2>&1
find "$out/prof_${tag}_bb" -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} "$dst/${tag}_kernel_stats_backbone.csv"
head -12 "$dst/${tag}_kernel_stats_b8.csv" | cut -c1-160
echo "== PMC passes"
cd "$root"
bash tools/gpu_pmc.sh "$tag" 2>&1 | tail -8
python tools/pmc_summary.py "$tag" "$dst/${tag}_pmc_traffic.json" "$commit" > "$dst/${tag}_pmc_summary.txt" 2>&1
head -30 "$dst/${tag}_pmc_summary.txt" | cut -c1-200
cat "$dst/${tag}_pmc_traffic.json"
echo "== backbone fused kernels: phase ablations + PMC"
bash tools/gpu_r06_bbpmc.sh "$tag" > "$dst/${tag}_backbone_fused_ablate_pmc.txt" 2>&1; tail -5 "$dst/${tag}_backbone_fused_ablate_pmc.txt" | cut -c1-200
echo "== SQ passes (conv kernels)"
bash tools/gpu_pmc_sq.sh "$tag" > "$dst/${tag}_pmc_sq_conv.txt" 2>&1; tail -4 "$dst/${tag}_pmc_sq_conv.txt" | cut -c1-200
