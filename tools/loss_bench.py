"""Micro-benchmark of the fused warp+photometric kernels (GPU box): algorithmic GB/s per SURVEY.md section 8d."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
from dvf import lib as L
from dvf.ops import PhotoLossFn, SmoothLossFn
from dvf.synthetic import synthetic_batch
for (b, c, h, w, feat) in [(4, 3, 256, 832, False), (8, 3, 384, 1280, False), (4, 32, 256, 832, True)]:
    batch = synthetic_batch(b, h, w, device="cuda")
    gen = torch.Generator().manual_seed(0)
    if c == 3:
        tgt, s0, s1 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
    else:
        tgt, s0, s1 = (torch.rand(b, c, h, w, generator=gen).cuda().requires_grad_(True) for _ in range(3))
    depth = (torch.rand(b, h, w, generator=gen) * 20 + 2).cuda().requires_grad_(True)
    pose = torch.zeros(2, b, 6, device="cuda"); pose[1, :, 0] = -0.54; pose[0] += 0.01
    pose.requires_grad_(True)
    for it in range(3):
        l = PhotoLossFn.apply(tgt, depth, pose, batch["K"], batch["Kinv"], None, 0, s0, s1); l.backward()
    torch.cuda.synchronize()
    L.TIMER = L.KernelTimer()
    for it in range(20):
        l = PhotoLossFn.apply(tgt, depth, pose, batch["K"], batch["Kinv"], None, 0, s0, s1); l.backward()
    summ = L.TIMER.summary(); L.TIMER = None
    line = f"B{b} C{c} {h}x{w}:"
    for k in ("photo_fwd", "photo_bwd"):
        d = summ[k]; ms = d["ms"] / d["calls"]
        line += f"  {k} {ms*1e3:7.1f} us {d['bytes']/d['calls']/(ms*1e-3)/1e9:7.0f} GB/s ({d['bytes']/d['calls']/1e6:.0f} MB)"
    print(line, flush=True)
