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
