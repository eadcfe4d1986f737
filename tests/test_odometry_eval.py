"""Known-answer tests of the trajectory writer and the KITTI odometry metric.  The reference's tool
(tools/evaluation_tools.py) needs caffe / h5py / cv2 and cannot be imported: parity unpinned, algorithm restated."""
import math
import os

import numpy as np

import odometry_eval as oe


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]])


def _straight(n, step):
    poses = {}
    for i in range(n):
        P = np.eye(4)
        P[2, 3] = i * step
        poses[i] = P
    return poses


def test_pose_file_round_trip(tmp_path):
    rel = [_rot_y(0.01 * (i + 1)) @ np.array([[1, 0, 0, 0.1 * i], [0, 1, 0, 0], [0, 0, 1, 1.0], [0, 0, 0, 1.0]]) for i in range(5)]
    world = oe.se3_cam2world(rel)
    assert len(world) == 6 and np.allclose(world[0], np.eye(4))
    assert np.allclose(world[2], rel[0] @ rel[1])
    for p in world:
        oe.save_result_poses(p, str(tmp_path), "pred.txt")
    lines = open(os.path.join(tmp_path, "pred.txt")).read().strip().split("\n")
    assert len(lines) == 6 and all(len(l.split(" ")) == 12 for l in lines)
    back = oe.load_poses(os.path.join(tmp_path, "pred.txt"))
    assert sorted(back) == list(range(6))
    for i, p in enumerate(world):
        assert np.allclose(back[i], p, atol=1e-12)
    # 13-number form: leading frame index
    with open(os.path.join(tmp_path, "idx.txt"), "w") as f:
        f.write("7 " + " ".join(str(v) for v in world[3][:3].reshape(12)) + "\n")
    assert list(oe.load_poses(os.path.join(tmp_path, "idx.txt"))) == [7]


def test_identical_trajectories_have_zero_error():
    gt = _straight(1200, 1.0)
    err = oe.sequence_errors(gt, gt)
    assert err and all(abs(r[1]) < 1e-12 and abs(r[2]) < 1e-12 for r in err)
    assert {r[3] for r in err} == set(oe.SEGMENT_LENGTHS)
    assert err[0][0] == 9 and err[0][3] == 100
    # speed: 1 m per frame at 10 Hz -> segment of 100 m ends at the first frame beyond it (102 frames)
    assert abs(err[0][4] - 100 / (0.1 * 102)) < 1e-12
    assert oe.overall_error(err) == (0.0, 0.0)
    assert np.allclose(oe.trajectory_distances(gt), np.arange(1200.0))


def test_scale_drift_gives_the_analytic_translation_error():
    gt = _straight(400, 1.0)
    res = _straight(400, 1.1)                          # 10 % scale error, no rotation error
    err = oe.sequence_errors(gt, res)
    assert err
    for first, r_err, t_err, length, speed in err:
        frames = next(i for i in range(first, 400) if i - first > length) - first       # frames in the segment
        assert abs(r_err) < 1e-12
        assert abs(t_err - 0.1 * frames / length) < 1e-9
    t, r = oe.overall_error(err)
    assert 0.1 < t < 0.11 and r == 0.0


def test_constant_yaw_drift_gives_the_analytic_rotation_error():
    n, yaw = 300, 0.002
    gt = _straight(n, 1.0)
    res = {i: _rot_y(yaw * i) @ gt[i] for i in range(n)}     # orientation drifts linearly with the frame index
    err = oe.sequence_errors(gt, res)
    assert err
    for first, r_err, t_err, length, speed in err:
        frames = next(i for i in range(first, n) if i - first > length) - first
        assert abs(r_err * length - yaw * frames) < 1e-9


def test_compute_pose_error_known_answers():
    gt = np.stack([np.eye(4)[:3]] * 4)
    pred = gt.copy()
    assert oe.compute_pose_error(gt, pred) == (0.0, 0.0)
    pred[:, 0, 3] += 0.5                                # 0.5 m offset on every pose of the snippet
    ate, re = oe.compute_pose_error(gt, pred)
    assert abs(ate - np.linalg.norm([0.5] * 4) / 4) < 1e-12 and re == 0.0
    rot = np.stack([_rot_y(0.1)[:3]] * 4)
    ate, re = oe.compute_pose_error(gt, rot)
    assert abs(re - 0.1) < 1e-12
