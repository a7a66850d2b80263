cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
echo "=== bench default"; python bench.py --no-cpu-baseline
echo "=== smoke"; python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
) > gpurun_out/diag25.log 2>&1
