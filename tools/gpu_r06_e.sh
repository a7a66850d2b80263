#!/usr/bin/env bash
root="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$root"; export TMPDIR=/tmp
FVP_TEST_DIAG_LIB=1 FVP_WINO_GENERIC=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15
cp gpurun_out/parity_report.jsonl gpurun_out/parity_report_wino_generic.jsonl
