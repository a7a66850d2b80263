#!/usr/bin/env bash
# round 6: backbone fused kernels - phase ablations (diagnostics build) and PMC passes (traffic, MFMA busy, LDS)
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"; export TMPDIR=/tmp
tag="${1:-r06}"
for ab in 0 1 2 3 4 7; do
  echo "== FVP_BB_ABLATE=$ab (1 no residual loads, 2 no stores, 4 x from one pixel)"
  FVP_TEST_DIAG_LIB=1 FVP_LIB="$root/tests/diag/libfvp_hip_diag.so" FVP_BB_ABLATE=$ab timeout 300 python tools/bench_backbone.py --images 40 --iters 3 --per-op 2>&1 | grep -E "fused group|ms/pass"
done
cd /tmp
cmd="python $root/tools/bench_backbone.py --images 40 --iters 2"
pm="$out/pmc_bb_${tag}"; rm -rf "$pm"; mkdir -p "$pm"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $pm/$name -o p -- $cmd > $pm/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA
run tcc1 TCC_HIT TCC_MISS TCC_READ TCC_EA0_RDREQ
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $pm/stats -o trace -- $cmd > $pm/stats.log 2>&1
python3 - "$pm" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
dur = {}
for path in glob.glob(root + "/stats/**/*kernel_stats*.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Name"].split("(")[0][-44:]] = float(r["AverageNs"]) / 1e3
for path in sorted(glob.glob(root + "/*/**/p_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "k_bb_" not in n or "pack" in n: continue
        key = n.split("(")[0][-44:]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (r["Dispatch_Id"], key)
        if did not in seen: seen.add(did); cnt[key] += 1
    print("==", path.split("/")[-3] if "pmc" not in path.split("/")[-2] else path.split("/")[-2])
    for k in sorted(acc):
        print(f"{k:46s} n={cnt[k]:3d} avg_us={dur.get(k, 0):8.1f} " + " ".join(f"{c}={v / cnt[k]:.4g}" for c, v in sorted(acc[k].items())))
PY
