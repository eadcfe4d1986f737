"""Pose-trajectory output and KITTI odometry evaluation: the host-side utilities either side of the
training hot path, rewritten with vectorised numpy.

  save_result_poses / se3_cam2world / compute_pose_error   reference pytorch_version/inference.py:43-88
  load_poses / trajectory_distances / sequence_errors / overall_error   reference tools/evaluation_tools.py:384-508
                                                             (class kittiEvalOdom, the KITTI devkit metric)
The reference tool cannot be imported here (caffe, h5py, cv2), so these follow its published algorithm and are
checked by known-answer tests (tests/test_odometry_eval.py): parity unpinned against the reference itself."""
import os

import numpy as np

SEGMENT_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)     # metres (evaluation_tools.py:391)
FIRST_FRAME, FRAME_STEP, FRAME_PERIOD = 9, 10, 0.1             # :456,:458 and the 10 Hz speed constant of :482


def save_result_poses(se3, output_dir, filename):
    """Append one pose as a KITTI line: the 3x4 [R|t] in row-major order, 12 numbers (inference.py:43-61)."""
    se3 = np.asarray(se3, dtype=np.float64)
    with open(os.path.join(output_dir, filename), "a") as f:
        f.write(" ".join(str(v) for v in se3[:3, :4].reshape(12)) + "\n")


def se3_cam2world(rel_poses):
    """Chain frame-to-frame [4,4] poses into camera-to-world poses, identity first (inference.py:63-70)."""
    out = [np.eye(4)]
    for p in rel_poses:
        out.append(out[-1] @ np.asarray(p, dtype=np.float64))
    return out


def compute_pose_error(gt, pred):
    """ATE and mean rotation error of a snippet of [N,3,4] poses (inference.py:72-88)."""
    gt, pred = np.asarray(gt, dtype=np.float64), np.asarray(pred, dtype=np.float64)
    n = gt.shape[0]
    ate = np.linalg.norm((gt[:, :, -1] - pred[:, :, -1]).reshape(-1))
    R = gt[:, :, :3] @ np.linalg.inv(pred[:, :, :3])
    s = np.sqrt((R[:, 0, 1] - R[:, 1, 0]) ** 2 + (R[:, 1, 2] - R[:, 2, 1]) ** 2 + (R[:, 0, 2] - R[:, 2, 0]) ** 2)
    c = np.trace(R, axis1=1, axis2=2) - 1.0
    return ate / n, float(np.arctan2(s, c).sum()) / n


def load_poses(file_name):
    """{frame index: [4,4]} from lines of 12 numbers, or 13 with a leading frame index (evaluation_tools.py:394-417)."""
    poses = {}
    with open(file_name) as f:
        for cnt, line in enumerate(f):
            vals = [float(v) for v in line.split()]
            if not vals:
                continue
            with_idx = len(vals) == 13
            P = np.eye(4)
            P[:3, :4] = np.asarray(vals[with_idx:with_idx + 12]).reshape(3, 4)
            poses[int(vals[0]) if with_idx else cnt] = P
    return poses


def trajectory_distances(poses):
    """Cumulative path length along the sorted frames (evaluation_tools.py:419-435)."""
    keys = sorted(poses)
    t = np.stack([poses[k][:3, 3] for k in keys])
    return np.concatenate(([0.0], np.cumsum(np.linalg.norm(np.diff(t, axis=0), axis=1))))


def sequence_errors(poses_gt, poses_result):
    """Rows [first_frame, rot_err/len (rad/m), trans_err/len, len, speed] for every start frame 9, 19, ... and every
    segment length whose end exists in both trajectories (evaluation_tools.py:452-485).  Frames are positions in the
    sorted ground-truth list, as in the reference (its dict keys are 0..N-1)."""
    keys = sorted(poses_gt)
    dist = trajectory_distances(poses_gt)
    err = []
    for first in range(FIRST_FRAME, len(keys), FRAME_STEP):
        for length in SEGMENT_LENGTHS:
            beyond = np.nonzero(dist[first:] > dist[first] + length)[0]
            if beyond.size == 0:
                continue
            last = first + int(beyond[0])
            if first not in poses_result or last not in poses_result:
                continue
            d_gt = np.linalg.inv(poses_gt[first]) @ poses_gt[last]
            d_res = np.linalg.inv(poses_result[first]) @ poses_result[last]
            e = np.linalg.inv(d_res) @ d_gt
            r_err = float(np.arccos(np.clip(0.5 * (np.trace(e[:3, :3]) - 1.0), -1.0, 1.0)))
            t_err = float(np.linalg.norm(e[:3, 3]))
            speed = length / (FRAME_PERIOD * (last - first + 1.0))
            err.append([first, r_err / length, t_err / length, length, speed])
    return err


def overall_error(seq_err):
    """(average translational error, average rotational error) over the rows of sequence_errors (:494-508)."""
    if not seq_err:
        return float("nan"), float("nan")
    a = np.asarray(seq_err, dtype=np.float64)
    return float(a[:, 2].mean()), float(a[:, 1].mean())
