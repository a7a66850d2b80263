cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
for st in 1 2 3 2; do
echo "=== e2e b8 s$st"; timeout 300 python bench.py --backbone --steps 8 --warmup 3 --no-cpu-baseline --streams $st | cut -c80-130
done
) > gpurun_out/diag58.log 2>&1
