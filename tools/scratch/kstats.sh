cd /tmp; export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/kstats; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $root/tools/scratch/run_step.py > $out/log 2>&1
f=$(find $out -name "*kernel_stats*.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    n=r["Name"].split("(")[0].replace("void ","")[:60]
    print(f"{n:62s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
