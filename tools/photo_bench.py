"""Micro-benchmark of the fused warp + photometric loss kernels (GPU box): per-launch time (HIP events on the launch
stream, median of N) and algorithmic HBM bandwidth (SURVEY.md section 8d byte model) for the benchmark's cases.
usage: photo_bench.py [iters]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
from dvf.ops import PhotoLossFn
from dvf import lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
flt = sys.argv[2] if len(sys.argv) > 2 else ""
CASES = [  # name, B, C, H, W, V, masks, grads to images
    ("cfg2 image  C=3  V=2 256x832  B=4", 4, 3, 256, 832, 2, False, False),
    ("cfg3 feat   C=32 V=2 256x832  B=4", 4, 32, 256, 832, 2, False, True),
    ("cfg3 feat   C=32 V=2 256x832  B=8", 8, 32, 256, 832, 2, False, True),
    ("cfg4 image  C=3  V=2 256x832  B=4 masks", 4, 3, 256, 832, 2, True, False),
    ("cfg5 image  C=3  V=4 384x1280 B=8 masks", 8, 3, 384, 1280, 4, True, False),
    ("cfg5 scale3 C=3  V=4 48x160   B=8 masks", 8, 3, 48, 160, 4, True, False),
]
g = torch.Generator().manual_seed(0)
for name, B, C, H, W, V, masks, img_grads in CASES:
    if flt not in name:
        continue
    mk = lambda: torch.nn.functional.avg_pool2d(torch.rand(B, C, H + 4, W + 4, generator=g), 5, 1).cuda()
    tgt = mk().requires_grad_(img_grads)
    srcs = [mk().requires_grad_(img_grads) for _ in range(V)]
    # KITTI-like geometry: smooth depth 5..40 m, small ego-motion, one stereo view with a 0.54 m baseline
    yy = torch.linspace(0, 1, H).view(1, H, 1)
    depth = (40.0 - 33.0 * yy + 2.0 * torch.sin(torch.linspace(0, 12, W)).view(1, 1, W)).expand(B, H, W).contiguous().cuda().requires_grad_(True)
    pose = (torch.randn(V, B, 6, generator=g) * 0.01)
    pose[1 % V, :, 0] -= 0.54
    pose = pose.cuda().requires_grad_(True)
    K = torch.tensor([[0.58 * W, 0, 0.5 * W], [0, 1.92 * H, 0.5 * H], [0, 0, 1.0]]).expand(B, 3, 3).contiguous()
    Kinv = torch.inverse(K[0]).expand(B, 3, 3).contiguous().cuda()
    K = K.cuda()
    mask = (torch.rand(B, V, H, W, generator=g) * 0.9 + 0.05).cuda().requires_grad_(True) if masks else None
    L.TIMER = L.KernelTimer()
    for _ in range(iters + 3):
        for t in [tgt, depth, pose, mask] + srcs:
            if t is not None:
                t.grad = None
        loss = PhotoLossFn.apply(tgt, depth, pose, K, Kinv, mask, 0, *srcs)
        loss.backward()
    torch.cuda.synchronize()
    rec = L.TIMER.records
    L.TIMER = None
    out = {}
    for kind in ("photo_fwd", "photo_bwd"):
        ms = [a.elapsed_time(b) for k, a, b, fl, by, tag in rec if k == kind][3:]
        by = [by for k, a, b, fl, by, tag in rec if k == kind][0]
        med = statistics.median(ms)
        out[kind] = (med, by / (med * 1e-3) / 1e9, by)
    print(f"{name:44s} fwd {out['photo_fwd'][0]*1e3:8.1f} us {out['photo_fwd'][1]:7.0f} GB/s ({out['photo_fwd'][2]/1e6:6.1f} MB) | "
          f"bwd {out['photo_bwd'][0]*1e3:8.1f} us {out['photo_bwd'][1]:7.0f} GB/s ({out['photo_bwd'][2]/1e6:6.1f} MB)  loss {float(loss):.5f}", flush=True)
