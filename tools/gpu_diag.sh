cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/diag42.log 2>&1
