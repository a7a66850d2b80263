cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15; tail -2 gpurun_out/parity_report.jsonl) > gpurun_out/diag29.log 2>&1
