"""How the heatmap seeds / MIN_SCORE of the float-parity fixtures (flavour "c" in cases.py) were
chosen (build container; uses the CPU oracle only, the fixtures themselves are then generated from
the reference by make_golden.py).

    python tests/golden/find_conditioned.py panoptic 2 "6,5" 1-12

For every heatmap seed: run the oracle with all proposals valid, evaluate the joint net in fp32 and
fp64, and look for a MIN_SCORE such that every proposal above it (in every frame) has an
fp32-vs-fp64 joint difference <= LIMIT, each frame keeps >= 3 people, and the threshold sits in a
relative confidence gap >= 15 %.  Prints the candidates; the chosen one is written into cases.py by hand."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fvp_oracle as O  # noqa: E402
import fvp_synthetic as S  # noqa: E402
from cases import CONDITIONED  # noqa: E402

LIMIT = float(os.environ.get("FVP_FLOOR_LIMIT", "4.5e-4"))


def main():
    shape, B = sys.argv[1], int(sys.argv[2])
    people = [int(v) for v in sys.argv[3].split(",")]
    lo, hi = (int(v) for v in sys.argv[4].split("-"))
    wseed = int(sys.argv[5]) if len(sys.argv) > 5 else 7
    torch.set_num_threads(8)
    cfg = S.make_cfg(shape, min_score=-1.0)
    cams, seq = S.load_cameras(shape)
    rt = S.resize_transform(cfg)
    meta = {"seq": [seq] * B}
    orc = O.Oracle(cfg, S.fill_state_dict_conditioned(O.reference_state_dict_shapes(cfg), seed=wseed))
    for hs in range(lo, hi + 1):
        heat = S.heatmaps_people(cfg, cams, seq, B, people if B > 1 else people[0], seed=hs, **CONDITIONED)
        with torch.no_grad():
            _, _, centers, _ = orc.hdn(heat, meta, cams, rt)
            mask = centers[:, :, 3] >= 0
            f32, _ = orc.jln(meta, heat, centers.clone(), mask, cams, rt)
            f64, _ = orc.jln(meta, heat, centers.clone(), mask, cams, rt, net_dtype=torch.float64)
        d = (f32 - f64).norm(dim=-1).max(dim=2)[0].numpy()              # [B,N]
        conf = centers[:, :, 4].numpy()
        best = None
        for t in np.sort(conf.reshape(-1)):
            above = conf > t
            if above.sum(1).min() < 3 or d[above].max() > LIMIT:
                continue
            nxt = conf[above].min()
            gap = (nxt - t) / nxt
            if gap >= 0.15:
                best = (float(np.sqrt(t * nxt)), above.sum(1).tolist(), float(d[above].max()), gap)
                break
        print(f"hseed {hs}: " + ("-" if best is None else
                                 f"MIN_SCORE {best[0]:.5f} valid/frame {best[1]} floor max {best[2]:.2e} gap {best[3]:.0%}"),
              "| conf*1e3:floor*1e4 " + " | ".join(" ".join(f"{c * 1e3:.1f}:{e * 1e4:.0f}" for c, e in
                                                            sorted(zip(conf[b], d[b]), reverse=True)) for b in range(B)))


if __name__ == "__main__":
    main()
