"""Recipes of the golden cases (shared by the generator and the tests).

name -> (shape set, heatmap flavour, batch, people, heatmap seed, weight seed, MIN_SCORE override)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import fvp_synthetic as S  # noqa: E402

CASES = {
    "tiny_g_b2_all": ("tiny", "g", 2, 2, 3, 7, -1.0),
    "tiny_u_b3_thr": ("tiny", "u", 3, 0, 2, 7, 45.0),
    "campus_u_b2_all": ("campus", "u", 2, 0, 2, 7, -1.0),
    "panoptic_g_b1_all": ("panoptic", "g", 1, 4, 3, 7, -1.0),
    "panoptic_g_b2_thr": ("panoptic", "g", 2, 3, 5, 7, 17.9),
    "panoptic_u_b1_all": ("panoptic", "u", 1, 0, 2, 7, -1.0),
    "shelf_g_b1_all": ("shelf", "g", 1, 4, 3, 11, -1.0),
    # BASELINE.json configs[3] shape: 128x128x32 detection grid, jln128
    "panoptic128_g_b1_all": ("panoptic128", "g", 1, 4, 10, 7, -1.0),
}

# Float-parity fixtures (flavour "c": conditioned weights + people heatmaps, SURVEY.md section 7 hard
# part 1): the reference's own fp32-vs-fp64 noise floor on these is <= ~3e-4 mm, so the north-star bar
# |build - reference| <= 1e-3 mm is asserted on them WITHOUT a noise-floor escape.  Heatmap seed and
# MIN_SCORE were picked by tests/golden/find_conditioned.py (every proposal above the threshold is a
# person with single-mode joint maps, and the threshold sits in a >= 15 % confidence gap).
CONDITIONED = dict(sigma=2.5, region=1400.0, spacing=1400.0, joint_std=(120.0, 120.0, 250.0))
CASES.update({
    "panoptic_c_b2_thr": ("panoptic", "c", 2, [6, 5], 6, 7, 0.387),        # 4 + 4 valid people
    "shelf_c_b1_thr": ("shelf", "c", 1, 5, 3, 11, 0.435),                  # 3 valid people
    # Campus (3 views, coordinates up to 7.5 m where one fp32 ulp is 4.9e-4 mm): `find_conditioned.py campus 2 "3,3" 1-8`
    # with FVP_FLOOR_LIMIT=9e-4 finds NO seed whose persons have a reference fp32-vs-fp64 floor below 8e-4 mm (every person
    # of every seed: 0.8e-3 .. 1.8e-3 mm), so the 1e-3 mm bar against the reference's fp32 output is below the reference's
    # own reproducibility there.  Seed 5 is the one where ALL ten proposals are single-mode (floors 1.1e-3 .. 1.7e-3 mm);
    # the threshold sits in the 36 % confidence gap between 0.696 and 1.092.  Its asserted bar is CAMPUS_RULE in common.py.
    "campus_c_b2_thr": ("campus", "c", 2, [3, 3], 5, 7, 0.872),            # 5 + 4 valid proposals
})


def make_inputs(case, device="cpu"):
    shape, flavour, B, people, hseed, wseed, ms = CASES[case]
    cfg = S.make_cfg(shape, device=device, min_score=ms)
    cams, seq = S.load_cameras(shape)
    rt = S.resize_transform(cfg)
    if flavour == "c":
        heat = S.heatmaps_people(cfg, cams, seq, B, people, seed=hseed, **CONDITIONED)
    elif flavour == "g":
        heat = S.heatmaps_blobs(cfg, cams, seq, B, people=people, seed=hseed)
    else:
        heat = S.heatmaps_uniform(cfg, B, seed=hseed)
    meta = {"seq": [seq] * B}
    return cfg, cams, seq, rt, heat, meta, wseed


def make_weights(case, state_dict_like):
    """Seeded weights of a case for every key of ``state_dict_like`` (a model's state_dict or the
    oracle's key -> zeros map): the conditioned recipe for flavour "c", the generic one otherwise."""
    flavour, wseed = CASES[case][1], CASES[case][5]
    if flavour == "c":
        return S.fill_state_dict_conditioned(state_dict_like, seed=wseed)
    return S.fill_state_dict(state_dict_like, seed=wseed)
