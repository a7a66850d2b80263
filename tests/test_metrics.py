"""Evaluators against numbers produced by the reference's own
``Panoptic.evaluate`` / ``Shelf.evaluate`` / ``Shelf.coco2shelf3D`` (tests/golden/metrics.npz,
generator make_golden_metrics.py; inputs are stored in the fixture).  CPU only."""
import os

import numpy as np

from faster_voxelpose_amd.core import metrics as M

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz"))


def test_panoptic_ap_recall_mpjpe_match_reference():
    counts = G["pan_gt_count"]
    ofs = np.concatenate([[0], np.cumsum(counts)])
    gts = [G["pan_gt"][a:b] for a, b in zip(ofs[:-1], ofs[1:])]
    vis = [G["pan_vis"][a:b] for a, b in zip(ofs[:-1], ofs[1:])]
    r = M.evaluate_panoptic(list(G["pan_preds"]), gts, vis)
    assert abs(r["metric"] - float(G["pan_metric"])) < 1e-12
    # the reference reports these through a formatted message (4 / 3 decimals)
    want = G["pan_numbers"]
    got = [r[f"ap@{t}"] for t in M.AP_THRESHOLDS_MM] + [r["recall"], r["mpjpe"]]
    np.testing.assert_allclose(got[:7], want[:7], atol=5.1e-5)
    assert abs(got[7] - want[7]) < 5.1e-4


def test_coco_to_shelf_matches_reference():
    got = np.stack([M.coco_to_shelf(p[0, :, :3]) for p in G["shelf_preds"]])
    np.testing.assert_allclose(got, G["shelf_conv"], rtol=0, atol=1e-9)


def test_pcp_matches_reference():
    actors = [[None if np.isnan(a).any() else a * 1000.0 for a in row] for row in G["shelf_actors"]]
    r = M.evaluate_pcp(list(G["shelf_preds"]), actors)
    assert abs(r["metric"] - float(G["shelf_metric"])) < 1e-12
    want = G["shelf_numbers"]                              # actor 1..3 PCP %, average %, ..., recall (last)
    np.testing.assert_allclose(r["actor_pcp"][:3] * 100, want[:3], atol=5.1e-3)
    assert abs(r["avg_pcp"] * 100 - want[3]) < 5.1e-3 and abs(r["recall"] - want[-1]) < 5.1e-5
    assert set(r["bone_group_pcp"]) == {"Head", "Torso", "Upper arms", "Lower arms", "Upper legs", "Lower legs"}


def test_edge_cases():
    # no detections at all, frames without ground truth
    preds = [np.full((3, 15, 5), -1.0), np.full((3, 15, 5), -1.0)]
    gts = [np.zeros((0, 15, 3)), np.random.default_rng(0).normal(0, 100, (2, 15, 3))]
    vis = [np.zeros((0, 15)), np.ones((2, 15))]
    r = M.evaluate_panoptic(preds, gts, vis)
    assert r["metric"] == 0.0 and r["recall"] == 0.0 and r["mpjpe"] == float("inf")
    # a perfect detection
    p = np.full((3, 15, 5), -1.0)
    p[0, :, :3] = gts[1][1]
    p[0, :, 3] = 0
    p[0, :, 4] = 0.9
    r = M.evaluate_panoptic([preds[0], p], gts, vis)
    assert r["mpjpe"] == 0.0 and abs(r["recall"] - 0.5) < 1e-12 and r["ap@25"] > 0.49
