"""Diagnostic (GPU box): coordinate accuracy of the HIP warp vs the fp64 oracle."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
from oracle import geometry as og, losses as ol
import inverse_warp as iw, loss_functions as lf
import torch.nn.functional as F
b, c, h, w = 1, 3, 128, 416
gen = torch.Generator().manual_seed(b * 1000 + c * 100 + h)
R2, R1, L2 = (torch.rand(b, c, h, w, generator=gen) for _ in range(3))
depth = torch.rand(b, h, w, generator=gen) * 20 + 2
T21 = torch.randn(b, 6, generator=gen) * 0.03
TRL = torch.tensor([-0.54, 0, 0, 0, 0, 0.0]).expand(b, 6) + torch.randn(b, 6, generator=gen) * 0.005
K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
Kinv = torch.inverse(K[0]).expand(b, 3, 3).contiguous()
yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
ramp = torch.stack((xx, yy, torch.ones_like(xx)))[None].contiguous()
for pose, nm in ((T21, "T21"), (TRL, "TRL")):
    o64 = og.inverse_warp(ramp.double(), depth.double(), pose.double(), K.double(), Kinv.double())
    o32 = og.inverse_warp(ramp, depth, pose, K, Kinv)
    og_ = iw.inverse_warp(ramp.cuda(), depth.cuda(), pose.cuda(), K.cuda(), Kinv.cuda()).cpu()
    inter = (o64[:, 2] > 0.999)          # all four taps in bounds
    for k, lab in ((0, "x"), (1, "y")):
        e32 = (o32[:, k].double() - o64[:, k]).abs()[inter]
        eg = (og_[:, k].double() - o64[:, k]).abs()[inter]
        print(nm, lab, "oracle32 max/mean err px", e32.max().item(), e32.mean().item(), " hip max/mean", eg.max().item(), eg.mean().item())
for smooth in (False, True):
    imgs = [R2, R1, L2]
    if smooth:
        imgs = [F.avg_pool2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), 5, stride=1) for x in imgs]
    res = {}
    for tag, dt, dev in (("o64", torch.float64, "cpu"), ("o32", torch.float32, "cpu"), ("hip", torch.float32, "cuda")):
        xs = [x.detach().clone().to(dt).to(dev) for x in imgs + [depth, T21, TRL, K, Kinv]]
        xs[3].requires_grad_(True); xs[4].requires_grad_(True); xs[5].requires_grad_(True)
        fn = lf.photometric_reconstruction_loss if tag == "hip" else ol.photometric_reconstruction_loss
        l = fn(*xs); l.backward()
        res[tag] = (l.item(), xs[4].grad.double().cpu(), xs[3].grad.double().cpu())
    for tag in ("o32", "hip"):
        l, gp, gd = res[tag]; l0, gp0, gd0 = res["o64"]
        print("smooth" if smooth else "noise", tag, "loss rel", abs(l - l0) / l0, "pose rel", ((gp - gp0).abs().max() / gp0.abs().max()).item(),
              "depth outliers 1e-4/1e-3/1e-2", [int(((gd - gd0).abs() > t * gd0.abs().max()).sum()) for t in (1e-4, 1e-3, 1e-2)])
print("--- linear-ramp sources (no gradient discontinuity at tap-set crossings)")
gen = torch.Generator().manual_seed(77)
tgt = torch.rand(b, c, h, w, generator=gen) * 300
mk = lambda a_, b_, c_: (a_ * xx + b_ * yy + c_)
s1 = torch.stack((mk(0.7, 0.2, 5), mk(-0.3, 0.9, 60), mk(0.5, -0.4, 90)))[None].contiguous()
s2 = torch.stack((mk(0.1, 0.8, 15), mk(0.6, 0.3, 6), mk(-0.2, 0.5, 190)))[None].contiguous()
res = {}
for tag, dt, dev in (("o64", torch.float64, "cpu"), ("o32", torch.float32, "cpu"), ("hip", torch.float32, "cuda")):
    xs = [x.detach().clone().to(dt).to(dev) for x in [tgt, s1, s2, depth, T21, TRL, K, Kinv]]
    xs[3].requires_grad_(True); xs[4].requires_grad_(True); xs[5].requires_grad_(True)
    fn = lf.photometric_reconstruction_loss if tag == "hip" else ol.photometric_reconstruction_loss
    l = fn(*xs); l.backward()
    res[tag] = (l.item(), xs[4].grad.double().cpu(), xs[5].grad.double().cpu(), xs[3].grad.double().cpu())
for tag in ("o32", "hip"):
    l, gp, gq, gd = res[tag]; l0, gp0, gq0, gd0 = res["o64"]
    print(tag, "loss rel", abs(l - l0) / l0, "T21 rel", ((gp - gp0).abs().max() / gp0.abs().max()).item(),
          "TRL rel", ((gq - gq0).abs().max() / gq0.abs().max()).item(),
          "depth outliers", [int(((gd - gd0).abs() > t * gd0.abs().max()).sum()) for t in (1e-4, 1e-3, 1e-2)])
print("o64 T21", res["o64"][1]); print("hip T21", res["hip"][1]); print("o32 T21", res["o32"][1])
