"""End-to-end run of an entry script on real files: un_dataset.dataset -> DataLoader -> static input tensors -> HIP
training steps -> checkpoint in the reference's file names."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, PKG
from test_un_dataset import _make_tree

pytestmark = pytest.mark.gpu


def test_unsupervise_on_dataset_files(tmp_path):
    root = _make_tree(tmp_path, n=5)
    out = tmp_path / "ckpt"
    cmd = [sys.executable, os.path.join(PKG, "unsupervise.py"), "--data-root", str(root), "--epochs", "2", "-b", "2",
           "--height", "64", "--width", "128", "--output-dir", str(out), "--no-graph"]
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("Train epoch")]
    assert len(lines) == 2 and "samples/s" in lines[0]
    for l in lines:
        total = float(l.split("total:")[1].split()[0])
        assert total == total and 0 < total < 1e6          # finite, positive
    for name in ("best_vo_checkpoint.pth.tar", "best_depth_checkpoint.pth.tar"):
        sd = torch.load(os.path.join(out, name), map_location="cpu", weights_only=True)
        assert isinstance(sd, dict) and len(sd) > 10


def test_unsupervise_dvo_graph_mode_on_dataset_files(tmp_path):
    """unsupervise_dvo.py on dataset files through the default HIP-graph path: the stereo pose must reach the kernel in
    se(3) order as the files hold it (a swapped convention turns the 0.54 m baseline into a 0.54 rad rotation: the
    photometric term then collapses to the all-out-of-view value), the first real batch must be in place before the
    warm-up / capture steps, and checkpoints hold one network each (not the whole arena)."""
    root = _make_tree(tmp_path, n=5)
    out = tmp_path / "ckpt"
    cmd = [sys.executable, os.path.join(PKG, "unsupervise_dvo.py"), "--data-root", str(root), "--epochs", "2", "-b", "2",
           "--height", "64", "--width", "128", "--output-dir", str(out), "--log-interval", "1"]
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("Train epoch")]
    assert len(lines) == 2
    assert any(l.strip().startswith("epoch 0 [") for l in res.stdout.splitlines())         # --log-interval is honoured
    for l in lines:
        total = float(l.split("total:")[1].split()[0])
        assert total == total and 0 < total < 1e6
    vo = os.path.getsize(os.path.join(out, "best_vo_checkpoint.pth.tar"))
    depth = os.path.getsize(os.path.join(out, "best_depth_checkpoint.pth.tar"))
    assert vo < 20e6 < depth < 140e6, (vo, depth)         # 2.7 M vs 31.6 M parameters, fp32


def test_dvo_stereo_pose_convention_reaches_the_kernel():
    """The pose unsupervise_dvo.py hands to the loss for a dataset batch is (0, 0, 0, Tx, 0, 0): with it the stereo view
    of a fronto-parallel plane at depth Z is the left image shifted by fx * Tx / Z pixels (no rotation)."""
    import un_dataset
    from dvf import lib as L
    from dvf.ops import PhotoLossFn
    b, h, w, Z, tx = 1, 32, 128, 4.0, -0.1
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
    sample = [torch.zeros(b, 3, h, w)] * 3 + [K, torch.inverse(K[0]).expand(b, 3, 3).contiguous(), K,
                                              torch.tensor([[[0, 0, 0, tx, 0, 0.0]]])]
    batch = un_dataset.to_batch(sample, "cuda")
    shift = 0.58 * w * tx / Z                                             # -1.856 px
    xs = torch.arange(w, dtype=torch.float32)
    left = (xs * 0.01).expand(b, 3, h, w).contiguous().cuda()             # a ramp: sampling at x + shift gives an exact value
    right = ((xs + shift) * 0.01).expand(b, 3, h, w).contiguous().cuda()  # what the right camera must see
    depth = torch.full((b, h, w), Z, device="cuda")
    pose = batch["T_R2L_se3"].unsqueeze(0).contiguous()
    loss = PhotoLossFn.apply(right, depth, pose, batch["K"], batch["Kinv"], None, L.POSE_SE3 | L.PIXEL_COORDS, left)
    valid = 1.0 - 2.0 / w                                                 # columns whose source falls left of the image are masked
    assert float(loss) < 2e-4 * valid, float(loss)                        # matches up to rounding where in view
    swapped = batch["T_R2L_se3"][:, [3, 4, 5, 0, 1, 2]].unsqueeze(0).contiguous()
    bad = PhotoLossFn.apply(right, depth, swapped, batch["K"], batch["Kinv"], None, L.POSE_SE3 | L.PIXEL_COORDS, left)
    # (0.1 read as a rotation about x moves the view by ~6 rows and not at all along x: the ramp then mismatches by the
    # whole 1.86 px shift)
    assert float(bad) > 100 * max(float(loss), 1e-6), (float(bad), float(loss))     # the convention clash is not silent


def test_train_with_validate_on_odometry_tree(tmp_path):
    """train.py on dataset files with the reference's validate() (train.py:220-247) deciding the best checkpoint: pose
    network in eval mode on consecutive odometry frames, se(3) exponential map on the GPU, L1 against the relative pose."""
    from test_dataset_odometry import _make_odometry_tree
    root = _make_tree(tmp_path, n=5)
    (tmp_path / "odo").mkdir()
    val_root, _ = _make_odometry_tree(tmp_path / "odo")
    out = tmp_path / "ckpt"
    cmd = [sys.executable, os.path.join(PKG, "train.py"), "--data-root", str(root), "--val-root", str(val_root), "--epochs", "2",
           "-b", "2", "--height", "64", "--width", "128", "--output-dir", str(out), "-m", "0.2", "--test-sequences", "00"]
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    val = [l for l in res.stdout.splitlines() if l.startswith("Test set: Average loss:")]
    assert len(val) == 2
    v0 = float(val[0].split("loss:")[1].split()[0])
    assert v0 == v0 and 0 < v0 < 10 and "[BEST:True]" in val[0]
    assert os.path.exists(os.path.join(out, "best_vo_checkpoint.pth.tar"))
    train = [l for l in res.stdout.splitlines() if l.startswith("Train epoch")]
    assert len(train) == 2 and "exp:" in train[0]
