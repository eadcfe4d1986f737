"""Seeded KITTI-shaped synthetic batches: the metric is quoted on synthetic data because
the reference's datasets need KITTI on disk (un_dataset.py:21-22).  Same recipe as the test oracle's."""
import torch
import torch.nn.functional as F


def synthetic_batch(b, h, w, seed=1234, rank=0, n_views=2, smooth_images=True, device="cpu"):
    """dict(img_R2, img_R1, img_L2, extra_refs, K, Kinv, T_R2L): images U[0,255) (5x5 box low-pass so that
    photometric gradients are meaningful), K = [[.58W,0,.5W],[0,1.92H,.5H],[0,0,1]] (cf. data/dataset_builder.py:130-135),
    stereo pose (-0.54,0,0,0,0,0) in the (t, r) order of pose_vec2mat (inverse_warp.py:141-157).
    Tuple order of the reference dataset: (img_R1, img_L2, img_R2, K, Kinv, raw_K, T_R2L), un_dataset.py:78-84."""
    g = torch.Generator().manual_seed(seed + rank)
    imgs = []
    for _ in range(1 + n_views):
        im = torch.rand(b, 3, h, w, generator=g, dtype=torch.float64) * 255.0
        if smooth_images:
            im = F.avg_pool2d(F.pad(im, (2, 2, 2, 2), mode="reflect"), 5, stride=1)
        imgs.append(im.float())
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]], dtype=torch.float64)
    Kinv = torch.inverse(K)
    out = {"img_R2": imgs[0], "img_R1": imgs[1], "img_L2": imgs[2] if n_views >= 2 else imgs[1],
           "extra_refs": imgs[3:], "K": K.float().expand(b, 3, 3).contiguous(),
           "Kinv": Kinv.float().expand(b, 3, 3).contiguous(),
           "T_R2L": torch.tensor([-0.54, 0, 0, 0, 0, 0], dtype=torch.float32).expand(b, 6).contiguous(),
           # the same stereo pose as the reference dataset stores it: se(3) order (w, u)  (data/dataset_builder.py:155)
           "T_R2L_se3": torch.tensor([0, 0, 0, -0.54, 0, 0], dtype=torch.float32).expand(b, 6).contiguous()}
    return {k: ([t.to(device) for t in v] if isinstance(v, list) else v.to(device)) for k, v in out.items()}
