"""Depth metrics (reference tools/eval_depth_utils.py:10-28) against hand-computed known answers (the reference module
needs cv2 and cannot be imported here: parity pinned by these values)."""
import numpy as np

import depth_eval


def test_compute_errors_known_answers():
    gt = np.array([2.0, 4.0, 8.0, 10.0])
    pred = np.array([2.0, 5.0, 4.0, 20.0])
    abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 = depth_eval.compute_errors(gt, pred)
    assert np.isclose(abs_rel, (0 + 0.25 + 0.5 + 1.0) / 4)
    assert np.isclose(sq_rel, (0 + 1 / 4 + 16 / 8 + 100 / 10) / 4)
    assert np.isclose(rmse, np.sqrt((0 + 1 + 16 + 100) / 4))
    assert np.isclose(rmse_log, np.sqrt((0 + np.log(0.8) ** 2 + np.log(2) ** 2 + np.log(0.5) ** 2) / 4))
    # ratios: 1, 1.25, 2, 2 -> thresholds 1.25 (strict), 1.5625, 1.953125
    assert (a1, a2, a3) == (0.25, 0.5, 0.5)


def test_evaluate_protocol():
    rng = np.random.default_rng(0)
    gt = rng.uniform(0, 100, size=(40, 120))
    gt[rng.uniform(size=gt.shape) < 0.5] = 0.0                      # sparse LiDAR
    pred = gt * 0.5 + 1e-3                                          # a scaled prediction: median scaling recovers it
    res = depth_eval.evaluate([gt], [pred], garg_crop=True, median_scaling=True)
    assert res[0] < 1e-3 and res[4] == 1.0                          # abs_rel ~ 0, a1 = 1
    res2 = depth_eval.evaluate([gt], [pred], garg_crop=False, median_scaling=False)
    assert abs(res2[0] - 0.5) < 1e-3 and res2[4] == 0.0             # unscaled: 50 % relative error, ratio 2 everywhere
    m = depth_eval.garg_crop_mask(375, 1242)
    assert m.sum() == (int(0.99189189 * 375) - int(0.40810811 * 375)) * (int(0.96405229 * 1242) - int(0.03594771 * 1242))
