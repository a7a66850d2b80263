#!/usr/bin/env python
"""Rewrite tests/golden/parity_baseline.json from a parity report of the GPU suite:

    python tools/update_parity_baseline.py [gpurun_out/parity_report.jsonl] [commit]

The baseline pins what the build's float results are TODAY, fixture by fixture (max distance to the reference's fp32
output, to its float64 evaluation, the reference's own floor) and sweep by sweep.  The GPU suite fails when a number
drifts past the tolerances of tests/common.py::drift_check without this file changing in the same commit - a kernel that
re-orders a sum moves these numbers and must say so here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "parity_baseline.json")
FIX_KEYS = ("max_mm_vs_ref", "max_mm_vs_fp64", "ref_floor_mm", "worst_ratio_vs_fp64", "worst_ratio_vs_ref32")
SWEEP_KEYS = ("joints", "proposals", "max_mm_r1q", "max_mm_where_floor_le_4e-4", "violations_where_floor_le_4e-4",
              "worst_err_over_proposal_floor", "worst_proposal_err_over_own_floor", "proposals_within_2x_own_floor",
              "proposals_within_1.5x_own_floor", "frac_within_1e-3_mm")


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")
    commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT,
                                                                  capture_output=True, text=True).stdout.strip()
    fixtures, sweeps = {}, {}
    for line in open(src):
        d = json.loads(line)
        case = d.get("case", "")
        if case.startswith("sweep_"):
            sweeps[case[6:]] = {k: d[k] for k in SWEEP_KEYS if k in d}
        elif case:
            fixtures[case] = {k: d[k] for k in FIX_KEYS if k in d}
    out = {"_what": "float results of the HIP path on the MI355X per fixture / sweep; tests/common.py::drift_check compares every "
                    "GPU run with it (tools/update_parity_baseline.py rewrites it from a parity report)",
           "measured_at_commit": commit, "fixtures": fixtures, "sweeps": sweeps}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"wrote {os.path.relpath(OUT, ROOT)}: {len(fixtures)} fixtures, {len(sweeps)} sweeps (commit {commit})")


if __name__ == "__main__":
    main()
