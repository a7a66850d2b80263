"""Scoring of 3-D pose predictions -- numpy counterparts of the reference's dataset evaluators:

* Panoptic: greedy GT assignment by minimum MPJPE, AP at 25..150 mm, recall and MPJPE at 500 mm
  (``lib/dataset/panoptic.py`` ``evaluate`` :214-266, ``_eval_list_to_ap`` :268-297,
  ``_eval_list_to_mpjpe`` :299-311, ``_eval_list_to_recall`` :313-317);
* Shelf / Campus: percentage of correct parts on the 14-joint actor skeleton after the
  COCO-17 -> Shelf-14 conversion (``lib/dataset/shelf.py`` ``evaluate`` :162-227,
  ``coco2shelf3D`` :229-259).

Inputs are plain arrays (no dataset objects): ``preds[i]`` = ``[N, J, 5]`` rows of
``fused_poses`` for frame i (x, y, z, valid flag, confidence).
"""
import numpy as np

AP_THRESHOLDS_MM = (25, 50, 75, 100, 125, 150)


# ---- Panoptic ---------------------------------------------------------------------------------------
def match_to_ground_truth(preds, gt_joints, gt_vis):
    """Every valid predicted pose -> (mpjpe to its closest GT person, score, global GT id).
    ``gt_joints[i]`` = [P_i, J, 3], ``gt_vis[i]`` = [P_i, J]; frames without GT are skipped."""
    rows, gt_base = [], 0
    for pred, gts, vis in zip(preds, gt_joints, gt_vis):
        gts, vis = np.asarray(gts), np.asarray(vis)
        if len(gts) == 0:
            continue
        pred = np.asarray(pred)
        for pose in pred[pred[:, 0, 3] >= 0]:
            err = [np.mean(np.sqrt(np.sum((pose[v > 0.1, 0:3] - g[v > 0.1]) ** 2, axis=-1))) for g, v in zip(gts, vis)]
            k = int(np.argmin(err))
            rows.append((float(err[k]), float(pose[0, 4]), gt_base + k))
        gt_base += len(gts)
    return rows, gt_base


def _ranked(rows):
    # stable sort by descending score, like list.sort(key=score, reverse=True)
    return sorted(rows, key=lambda r: r[1], reverse=True)


def average_precision(rows, total_gt, threshold):
    """VOC-style AP of the ranked detections at one MPJPE threshold; also the final recall."""
    rows = _ranked(rows)
    hit = np.zeros(len(rows), dtype=bool)
    taken = set()
    for i, (err, _, gid) in enumerate(rows):
        if err < threshold and gid not in taken:
            hit[i] = True
            taken.add(gid)
    tp, fp = np.cumsum(hit), np.cumsum(~hit)
    recall = tp / (total_gt + 1e-5)
    precision = tp / (tp + fp + 1e-5)
    precision = np.maximum.accumulate(precision[::-1])[::-1] if len(rows) else precision   # monotone envelope
    precision = np.concatenate(([0], precision, [0]))
    recall = np.concatenate(([0], recall, [1]))
    step = np.where(recall[1:] != recall[:-1])[0]
    return float(np.sum((recall[step + 1] - recall[step]) * precision[step + 1])), float(recall[-2])


def matched_mpjpe(rows, threshold=500):
    taken, errs = set(), []
    for err, _, gid in _ranked(rows):
        if err < threshold and gid not in taken:
            errs.append(err)
            taken.add(gid)
    return float(np.mean(errs)) if errs else float("inf")


def recall_at(rows, total_gt, threshold=500):
    return len({gid for err, _, gid in rows if err < threshold}) / total_gt


def evaluate_panoptic(preds, gt_joints, gt_vis):
    rows, total_gt = match_to_ground_truth(preds, gt_joints, gt_vis)
    aps, recs = zip(*(average_precision(rows, total_gt, t) for t in AP_THRESHOLDS_MM))
    out = {f"ap@{t}": a for t, a in zip(AP_THRESHOLDS_MM, aps)}
    out.update(recall=recall_at(rows, total_gt), mpjpe=matched_mpjpe(rows), metric=float(np.mean(aps)),
               recall_per_threshold=list(recs))
    return out


# ---- Shelf / Campus ---------------------------------------------------------------------------------
_COCO_TO_SHELF = np.array([16, 14, 12, 11, 13, 15, 10, 8, 6, 5, 7, 9])
_LIMBS = ((0, 1), (1, 2), (3, 4), (4, 5), (6, 7), (7, 8), (9, 10), (10, 11), (12, 13))
BONE_GROUPS = {"Head": [8], "Torso": [9], "Upper arms": [5, 6], "Lower arms": [4, 7], "Upper legs": [1, 2],
               "Lower legs": [0, 3]}


def coco_to_shelf(coco_pose):
    """[17,3] COCO-order pose -> [14,3] Shelf order; neck / head top are interpolated from the
    shoulders, ears and nose (shelf.py:229-259)."""
    c = np.asarray(coco_pose, dtype=np.float64)
    s = np.zeros((14, 3))
    s[:12] = c[_COCO_TO_SHELF]
    mid_shoulder, head_center = (c[5] + c[6]) / 2, (c[3] + c[4]) / 2
    head_bottom = (mid_shoulder + head_center) / 2
    head_top = head_bottom + (head_center - head_bottom) * 2
    neck0 = (s[8] + s[9]) / 2
    top = neck0 + (c[0] - neck0) * np.array([0.75, 0.75, 1.5])
    neck = neck0 + (c[0] - neck0) * np.array([0.5, 0.5, 0.5])
    s[13] = top * 0.75 + head_top * (1 - 0.75)
    s[12] = neck * 0.75 + head_bottom * (1 - 0.75)
    return s


def evaluate_pcp(preds, actors_mm, alpha=0.5, recall_threshold=500):
    """``actors_mm[p][i]`` = [14,3] GT of actor p in frame i (millimetres) or None when absent."""
    P = len(actors_mm)
    correct, total = np.zeros(P), np.zeros(P)
    per_bone = np.zeros((P, 10))
    matched = seen = 0
    for i, pred in enumerate(preds):
        pred = np.asarray(pred)
        poses = np.stack([coco_to_shelf(p) for p in pred[pred[:, 0, 3] >= 0, :, :3]])
        for p in range(P):
            gt = actors_mm[p][i]
            if gt is None or len(gt) == 0:
                continue
            err = np.mean(np.sqrt(np.sum((gt[None] - poses) ** 2, axis=-1)), axis=-1)
            best = poses[int(np.argmin(err))]
            matched += bool(err.min() < recall_threshold)
            seen += 1
            hip_p, hip_g = (best[2] + best[3]) / 2.0, (gt[2] + gt[3]) / 2.0
            ends = [(best[a], best[b], gt[a], gt[b]) for a, b in _LIMBS] + [(hip_p, best[12], hip_g, gt[12])]
            for b, (pa, pb, ga, gb) in enumerate(ends):
                total[p] += 1
                if (np.linalg.norm(pa - ga) + np.linalg.norm(pb - gb)) / 2.0 <= alpha * np.linalg.norm(ga - gb):
                    correct[p] += 1
                    per_bone[p, b] += 1
    actor_pcp = correct / (total + 1e-8)
    groups = {k: np.sum(per_bone[:, v], axis=-1) / (total / 10 * len(v) + 1e-8) for k, v in BONE_GROUPS.items()}
    return dict(actor_pcp=actor_pcp, avg_pcp=float(np.mean(actor_pcp[:3])), recall=matched / (seen + 1e-8),
                bone_group_pcp=groups, metric=float(np.mean(actor_pcp[:3])))
