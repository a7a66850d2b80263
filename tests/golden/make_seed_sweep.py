"""Generate the seed-sweep float-parity fixtures from the reference itself (build container only).

    python tests/golden/make_seed_sweep.py [name ...]        # rewrites tests/golden/sweep_<name>.npz

For every sweep of seed_sweep.SWEEPS and every heatmap seed 1..10: the REFERENCE model (imported from
/root/reference/lib, _refimport.py) is run on the seeded inputs -> fused32 / centers / valid; the reference's own
rounding-noise floor per joint comes from the same joint net evaluated in float64 (the oracle's JLN with
net_dtype=float64 on the reference's proposals, exactly as make_golden.py does) -> fused64.  Data only; the
inputs are regenerated from the recipe on the GPU box.
"""
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
import fvp_oracle as O  # noqa: E402
import seed_sweep as SW  # noqa: E402


def run(ref, name):
    fused32, fused64, centers_all, valid_all = [], [], [], []
    model = orc = None
    for seed in SW.seeds_of(name):
        t0 = time.time()
        cfg, cams, seq, rt, heat, meta, _ = SW.make_inputs(name, seed)
        if model is None:
            with R.quiet():
                model = ref.faster_voxelpose.get(cfg).eval()
            sd = SW.make_weights(name, model.state_dict())
            model.load_state_dict(sd)
            orc = O.Oracle(cfg, sd)
        with torch.no_grad(), R.quiet():
            fused, _, centers, _, _ = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
            _, _, centers_hdn, _ = model.pose_net(heat, meta, cams, rt)
        mask = centers_hdn[:, :, 3] >= 0
        assert torch.equal(mask, centers[:, :, 3] >= 0)
        f64, _ = orc.jln(meta, heat, centers_hdn.clone(), mask, cams, rt, net_dtype=torch.float64)
        fused32.append(fused.numpy())
        fused64.append(f64.numpy())
        centers_all.append(centers.numpy())
        valid_all.append(mask.numpy())
        d = (fused[..., :3].double() - f64)[mask].norm(dim=-1)
        conf = centers[:, :, 4][mask]
        print(f"{name} seed {seed}: valid/frame {mask.sum(1).tolist()}  reference fp32-vs-fp64 per joint: max {float(d.max()) if d.numel() else 0:.2e} "
              f"median {float(d.median()) if d.numel() else 0:.2e}  joints with floor <= 4e-4: {int((d <= SW.FLOOR_OK).sum())}/{d.numel()}  "
              f"min |conf - thr| {float((conf - SW.MIN_SCORE).abs().min()) if conf.numel() else 0:.3f}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(SW.path(name), fused32=np.stack(fused32), fused64=np.stack(fused64), centers=np.stack(centers_all),
                        valid=np.stack(valid_all), seeds=np.array(SW.seeds_of(name), np.int64), min_score=np.float64(SW.MIN_SCORE))
    print(f"   -> {os.path.relpath(SW.path(name), ROOT)}  {os.path.getsize(SW.path(name)) / 1024:.0f} KiB", flush=True)


def main():
    ref = R.import_reference()
    torch.set_num_threads(int(os.environ.get("FVP_THREADS", "8")))
    for name in (sys.argv[1:] or list(SW.SWEEPS)):
        run(ref, name)


if __name__ == "__main__":
    main()
