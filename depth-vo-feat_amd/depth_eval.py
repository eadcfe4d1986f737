"""Depth evaluation metrics of the reference's tools (host side, numpy; not on the training hot path).

``compute_errors`` follows ``tools/eval_depth_utils.py:10-28`` (after Godard's monodepth evaluation): for ground-truth
and predicted depths of the valid pixels it returns (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3).  The reference module
imports cv2 (absent in the build image), so it cannot be imported for vectors: parity is pinned by hand-computed known
answers in tests/test_depth_eval.py.  ``evaluate`` applies the usual KITTI protocol of tools/eval_depth.py around it:
depth cap, optional Garg crop, optional median scaling."""
import numpy as np


def compute_errors(gt, pred):
    gt, pred = np.asarray(gt, dtype=np.float64), np.asarray(pred, dtype=np.float64)
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = ((thresh < 1.25 ** e).mean() for e in (1, 2, 3))
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def garg_crop_mask(h, w):
    """Crop of Garg et al. (ECCV16) used to reproduce Eigen's results (same constants as loss_functions_sfm.py:93-96)."""
    m = np.zeros((h, w), dtype=bool)
    m[int(0.40810811 * h):int(0.99189189 * h), int(0.03594771 * w):int(0.96405229 * w)] = True
    return m


def evaluate(gt_depths, pred_depths, min_depth=1e-3, max_depth=80.0, garg_crop=True, median_scaling=True):
    """Mean of compute_errors over a list of (gt, pred) depth maps of equal size per pair."""
    rows = []
    for gt, pred in zip(gt_depths, pred_depths):
        gt, pred = np.asarray(gt, dtype=np.float64), np.asarray(pred, dtype=np.float64)
        mask = (gt > min_depth) & (gt < max_depth)
        if garg_crop:
            mask &= garg_crop_mask(*gt.shape)
        g, p = gt[mask], np.clip(pred[mask], min_depth, max_depth)
        if median_scaling:
            p = p * (np.median(g) / np.median(p))
            p = np.clip(p, min_depth, max_depth)
        rows.append(compute_errors(g, p))
    return tuple(np.mean(np.array(rows), axis=0))
