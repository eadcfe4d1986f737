"""GPU input pipeline: dvf.image_ops.gpu_imresize against the host path of the loaders
(bytescale + PIL BILINEAR, the semantics of the reference's scipy.misc.imresize call) -- EXACT equality, including
KITTI's native frame size down to the benchmark size, a different drive size, and up-scaling."""
import numpy as np
import pytest
import torch

import un_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,dst", [((375, 1242), (256, 832)), ((370, 1226), (128, 416)), ((37, 123), (64, 128)),
                                     ((376, 1241), (384, 1280))])
def test_gpu_imresize_equals_host_path(src, dst):
    from dvf.image_ops import gpu_imresize
    rng = np.random.default_rng(src[0] + dst[0])
    img = rng.integers(7, 240, size=(src[0], src[1], 3), dtype=np.uint8)
    ref = un_dataset.imresize(img.astype(np.float32), dst).astype(np.float32).transpose(2, 0, 1)
    out = gpu_imresize(torch.from_numpy(img).cuda(), dst)
    assert out.dtype == torch.float32 and tuple(out.shape) == (3, dst[0], dst[1])
    assert np.array_equal(out.cpu().numpy(), ref)


def test_raw_loader_batch_equals_host_batch(tmp_path):
    from test_un_dataset import _make_tree
    root = _make_tree(tmp_path)
    host = un_dataset.dataset(img_height=64, img_width=128, root=str(root), shuffle=False)
    raw = un_dataset.dataset(img_height=64, img_width=128, root=str(root), shuffle=False, raw=True)
    hb = un_dataset.to_batch(next(iter(torch.utils.data.DataLoader(host, batch_size=2, shuffle=False))), "cuda")
    rb = un_dataset.to_batch_raw(next(iter(torch.utils.data.DataLoader(raw, batch_size=2, shuffle=False,
                                                                     collate_fn=un_dataset.collate_raw))), "cuda", (64, 128))
    for k in ("img_R1", "img_L2", "img_R2", "K", "Kinv", "T_R2L", "T_R2L_se3"):
        assert torch.equal(hb[k], rb[k]), k
