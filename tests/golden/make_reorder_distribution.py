"""How far does the REFERENCE land from ITSELF when only the summation order of its convolutions changes?  (build container only)

    python tests/golden/make_reorder_distribution.py [name ...]     # rewrites tests/golden/reorder_distribution.json

VERDICT round 5, item 4: the factors of the float-parity rules (seed_sweep.py R2: 3 x the proposal's floor; the Campus
floor rule of tests/common.py: 2 x / 3 x) were constants the builder chose.  This script measures the quantity they are
meant to bound on the reference itself: for every seed of every sweep the reference model (imported from
/root/reference/lib) is run twice on the same inputs and the same proposals -

    ref32   as the sweep fixtures were made: oneDNN convolutions, FVP_THREADS (8) threads  (== sweep_<name>.npz fused32)
    ref32'  the same modules with oneDNN switched off (torch.backends.mkldnn.flags(enabled=False): ATen's native
            im2col + GEMM convolution) on ONE thread - another summation order of the same fp32 products

- and the per-proposal ratio  max_j |ref32' - ref32| / pfloor  (pfloor = max_j |ref32 - ref64|, the fixture's own
fp32-vs-fp64 floor of that proposal) and the per-joint ratio  |ref32' - ref32| / max(pfloor, 4e-4)  are collected.
Their distribution is what "another implementation of the same arithmetic" looks like through the reference's own
nets; the test derives its factors from it (tests/golden/seed_sweep.py::measured_factors) instead of from constants.
Data only: summary statistics per sweep (no tensors)."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _refimport as R  # noqa: E402
import seed_sweep as SW  # noqa: E402

OUT = os.path.join(HERE, "reorder_distribution.json")


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else 0.0


def run(ref, name):
    fx = np.load(SW.path(name))
    prop_ratio, joint_ratio, joint_mm = [], [], []
    model = None
    for si, seed in enumerate(SW.seeds_of(name)):
        t0 = time.time()
        cfg, cams, seq, rt, heat, meta, _ = SW.make_inputs(name, seed)
        if model is None:
            with R.quiet():
                model = ref.faster_voxelpose.get(cfg).eval()
            model.load_state_dict(SW.make_weights(name, model.state_dict()))
        torch.set_num_threads(int(os.environ.get("FVP_THREADS", "8")))
        with torch.no_grad(), R.quiet():
            fused_a, _, centers_a, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        assert np.array_equal(fused_a.numpy(), fx["fused32"][si]), "the reference no longer reproduces its own fixture"
        torch.set_num_threads(1)
        with torch.no_grad(), R.quiet(), torch.backends.mkldnn.flags(enabled=False):
            fused_b, _, centers_b, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
        valid = fx["valid"][si]
        same = np.array_equal(centers_a.numpy()[..., :4], centers_b.numpy()[..., :4])
        d = np.linalg.norm(fused_b.numpy()[..., :3].astype(np.float64) - fused_a.numpy()[..., :3], axis=-1)      # [B,N,J]
        fl = np.linalg.norm(fx["fused32"][si][..., :3].astype(np.float64) - fx["fused64"][si], axis=-1)
        pfl = fl.max(axis=-1)                                                                                # [B,N]
        # the sweep's own exclusion: proposals whose confidence sits on the threshold
        conf = fx["centers"][si][..., 4]
        cmp = valid & (np.abs(conf - SW.MIN_SCORE) > SW.THRESHOLD_GUARD * SW.MIN_SCORE)
        if not same:
            cmp &= (centers_b.numpy()[..., 3] >= 0) & np.all(centers_a.numpy()[..., :3] == centers_b.numpy()[..., :3], axis=-1)
        prop_ratio += list((d.max(axis=-1)[cmp] / pfl[cmp]))
        joint_ratio += list((d[cmp] / np.maximum(pfl[cmp], SW.FLOOR_OK)[:, None]).ravel())
        joint_mm += list(d[cmp].ravel())
        print(f"{name} seed {seed}: proposals {int(cmp.sum())}  max |ref32' - ref32| {d[cmp].max():.2e} mm  "
              f"worst proposal ratio {float((d.max(axis=-1)[cmp] / pfl[cmp]).max()):.2f}  centres equal {same}  ({time.time() - t0:.0f} s)", flush=True)
    pr, jr, jm = np.array(prop_ratio), np.array(joint_ratio), np.array(joint_mm)
    return {"proposals": int(len(pr)), "joints": int(len(jr)),
            "proposal_ratio": {"p50": pct(pr, 50), "p90": pct(pr, 90), "p95": pct(pr, 95), "p99": pct(pr, 99), "max": float(pr.max())},
            "proposals_within_2x_own_floor": int((pr <= 2.0).sum()), "proposals_within_1.5x_own_floor": int((pr <= 1.5).sum()),
            "joint_ratio_r2": {"p50": pct(jr, 50), "p99": pct(jr, 99), "max": float(jr.max())},
            "joint_mm": {"p50": pct(jm, 50), "p99": pct(jm, 99), "max": float(jm.max())},
            "frac_joints_within_1e-3_mm": float((jm <= SW.BAR_MM).mean())}


def main():
    ref = R.import_reference()
    out = json.load(open(OUT)) if os.path.isfile(OUT) else {}
    out["_what"] = ("reference vs reference: oneDNN convs on 8 threads (the fixtures) against ATen's native convs on one thread; "
                    "ratios to the proposal's own fp32-vs-fp64 floor (make_reorder_distribution.py)")
    for name in (sys.argv[1:] or list(SW.SWEEPS)):
        out[name] = run(ref, name)
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
        print(name, json.dumps(out[name]), flush=True)


if __name__ == "__main__":
    main()
