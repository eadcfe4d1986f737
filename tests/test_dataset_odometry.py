"""dataset.pose_framework_KITTI (reference pytorch_version/dataset.py:11-97) on a synthetic KITTI-odometry-shaped tree:
pair construction, channel order and the relative ground-truth pose."""
import numpy as np
import torch
from PIL import Image

import dataset


def _make_odometry_tree(tmp_path, n=4):
    rng = np.random.default_rng(1)
    poses = []
    for seq in ("00", "03"):
        d = tmp_path / "sequences" / seq / "image_2"
        d.mkdir(parents=True)
        rows = []
        for i in range(n):
            Image.fromarray(rng.integers(10, 240, size=(20, 60, 3), dtype=np.uint8)).save(d / f"{i:06d}.png")
            ang = 0.05 * i
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            rows.append(np.concatenate([R, np.array([[0.1 * i], [0.0], [1.0 * i]])], axis=1).reshape(-1))
        (tmp_path / "poses").mkdir(exist_ok=True)
        np.savetxt(tmp_path / "poses" / f"{seq}.txt", np.array(rows))
        poses.append(np.array(rows).reshape(-1, 3, 4))
    return tmp_path, poses


def test_pairs_and_relative_pose(tmp_path):
    root, poses = _make_odometry_tree(tmp_path)
    ds = dataset.pose_framework_KITTI(str(root), ["00"], img_height=16, img_width=48, shuffle=False)
    assert len(ds) == 3 and ds.sequence_num == 1
    data, pose = ds[1]                                       # frames (1, 2)
    assert tuple(data.shape) == (6, 16, 48) and data.dtype == torch.float32 and tuple(pose.shape) == (4, 4)
    T1, T2 = np.eye(4), np.eye(4)
    T1[:3], T2[:3] = poses[0][1], poses[0][2]
    assert np.allclose(pose.numpy(), np.linalg.inv(T1) @ T2, atol=1e-6)
    # channels 0-2 are the LATER frame (reference dataset.py:60-61)
    later = dataset.imresize(dataset.imread(ds.samples[1]["imgs"][1]).astype(np.float32), (16, 48)).astype(np.float32)
    assert np.array_equal(data[:3].numpy(), later.transpose(2, 0, 1))
    both = dataset.pose_framework_KITTI(str(root), ["00", "03"], img_height=16, img_width=48)
    assert len(both) == 6 and both.sequence_num == 2
