"""Accuracy of single convolution layers against an fp64 torch-CPU run: rms error / rms value of forward, dgrad and wgrad
for the HIP kernels and for torch's own fp32 CPU kernels (the reference's arithmetic).  usage: layer_noise.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
import torch
import torch.nn.functional as F
from dvf.conv import ConvFn

LAYERS = [  # name, cin segs, cout, k, s, p, op, transposed, (N,H,W)
    ("conv3.2 128->128 @32x104", [128], 128, 3, 1, 1, 0, False, (4, 32, 104)),
    ("conv4.0 s2 128->256 32x104", [128], 256, 3, 2, 1, 0, False, (4, 32, 104)),
    ("conv4.2 256->256 @16x52", [256], 256, 3, 1, 1, 0, False, (4, 16, 52)),
    ("conv5.2 512->512 @8x26", [512], 512, 3, 1, 1, 0, False, (4, 8, 26)),
    ("conv6.2 512->512 @4x13", [512], 512, 3, 1, 1, 0, False, (4, 4, 13)),
    ("conv7.2 512->512 @2x7", [512], 512, 3, 1, 1, 0, False, (4, 2, 7)),
    ("upconv7 T 512->512 2x7", [512], 512, 3, 2, 1, 1, True, (4, 2, 7)),
    ("iconv7 1024->512 @4x13", [512, 512], 512, 3, 1, 1, 0, False, (4, 4, 13)),
    ("iconv5 512->256 @16x52", [256, 256], 256, 3, 1, 1, 0, False, (4, 16, 52)),
    ("upconv3 T 128->64 32x104", [128], 64, 3, 2, 1, 1, True, (4, 32, 104)),
    ("upconv4 T 256->128 16x52", [256], 128, 3, 2, 1, 1, True, (4, 16, 52)),
    ("iconv4 256->128 @32x104", [128, 128], 128, 3, 1, 1, 0, False, (4, 32, 104)),
    ("iconv3 129->64 @64x208", [64, 64, 1], 64, 3, 1, 1, 0, False, (4, 64, 208)),
    ("predict4 128->1 @32x104", [128], 1, 3, 1, 1, 0, False, (4, 32, 104)),
    ("predict3 64->1 @64x208", [64], 1, 3, 1, 1, 0, False, (4, 64, 208)),
]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if any(a in l[0] for a in sys.argv[1:])]
rms = lambda x: float(x.double().pow(2).mean().sqrt())
for name, segs, cout, k, s, p, op, tr, (n, h, w) in LAYERS:
    g = torch.Generator().manual_seed(5)
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, generator=g).relu() for c in segs]          # post-ReLU activations, as in the net
    wt = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        rx = [x.to(dt).detach().clone().requires_grad_(True) for x in xs]
        rw, rb = wt.to(dt).detach().clone().requires_grad_(True), b.to(dt).detach().clone().requires_grad_(True)
        xin = torch.cat(rx, 1)
        pre = F.conv_transpose2d(xin, rw, rb, stride=s, padding=p, output_padding=op) if tr else F.conv2d(xin, rw, rb, stride=s, padding=p)
        if tag == "f64":
            gout = torch.randn(pre.shape, generator=g, dtype=torch.float64)
        (pre * gout.to(dt)).sum().backward()
        res[tag] = (pre.detach(), [x.grad for x in rx], rw.grad)
    gx = [x.cuda().detach().clone().requires_grad_(True) for x in xs]
    gw, gb = wt.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    out = ConvFn.apply(gw, gb, (k, s, p, op, tr, 0, 1.0, 0.0, None), *gx)
    (out * gout.float().cuda()).sum().backward()
    res["hip"] = (out.detach().cpu(), [x.grad.cpu() for x in gx], gw.grad.cpu())
    ref = res["f64"]
    line = f"{name:28s}"
    for tag in ("hip", "f32"):
        r = res[tag]
        e_f = rms(r[0].double() - ref[0]) / rms(ref[0])
        e_d = max(rms(a.double() - c) / rms(c) for a, c in zip(r[1], ref[1]))
        e_w = rms(r[2].double() - ref[2]) / rms(ref[2])
        line += f" | {tag}: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e}"
    print(line, flush=True)
