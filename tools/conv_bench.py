"""Per-layer micro-benchmark of the convolution kernels (GPU box).  usage: conv_bench.py [filter] ; env DVF_DBG for
ablations (1: no patch staging, 2: no weight staging, 4: no MFMA loop)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
from dvf.conv import ConvFn
from dvf import lib as L
LAYERS = [  # name, cin segs, cout, k, s, p, op, transposed, act, (N,H,W), out_hw
    ("conv1.2 7x7s1 32->32 @128x416", [32], 32, 7, 1, 3, 0, False, 1, (4, 128, 416), None),
    ("conv2.0 5x5s2 32->64", [32], 64, 5, 2, 2, 0, False, 1, (4, 128, 416), None),
    ("conv2.2 5x5s1 64->64 @64x208", [64], 64, 5, 1, 2, 0, False, 1, (4, 64, 208), None),
    ("conv3.0 3x3s2 64->128", [64], 128, 3, 2, 1, 0, False, 1, (4, 64, 208), None),
    ("conv3.2 3x3s1 128->128 @32x104", [128], 128, 3, 1, 1, 0, False, 1, (4, 32, 104), None),
    ("conv4.2 3x3s1 256->256 @16x52", [256], 256, 3, 1, 1, 0, False, 1, (4, 16, 52), None),
    ("conv5.2 3x3s1 512->512 @8x26", [512], 512, 3, 1, 1, 0, False, 1, (4, 8, 26), None),
    ("conv6.2 3x3s1 512->512 @4x13", [512], 512, 3, 1, 1, 0, False, 1, (4, 4, 13), None),
    ("iconv4 3x3s1 256->128 @32x104", [128, 128], 128, 3, 1, 1, 0, False, 1, (4, 32, 104), None),
    ("iconv2 3x3s1 65->32 @128x416", [32, 32, 1], 32, 3, 1, 1, 0, False, 1, (4, 128, 416), None),
    ("iconv1 3x3s1 17->16 @256x832", [16, 1], 16, 3, 1, 1, 0, False, 1, (4, 256, 832), None),
    ("disp1 3x3s1 16->1 @256x832", [16], 1, 3, 1, 1, 0, False, 2, (4, 256, 832), None),
    ("disp2 3x3s1 32->1 @128x416", [32], 1, 3, 1, 1, 0, False, 2, (4, 128, 416), None),
    ("disp3 3x3s1 64->1 @64x208", [64], 1, 3, 1, 1, 0, False, 2, (4, 64, 208), None),
    ("disp4 3x3s1 128->1 @32x104", [128], 1, 3, 1, 1, 0, False, 2, (4, 32, 104), None),
    ("upconv4 T3x3s2 256->128 ->32x104", [256], 128, 3, 2, 1, 1, True, 1, (4, 16, 52), None),
    ("upconv2 T3x3s2 64->32 ->128x416", [64], 32, 3, 2, 1, 1, True, 1, (4, 64, 208), None),
    ("upconv1 T3x3s2 32->16 ->256x832", [32], 16, 3, 2, 1, 1, True, 1, (4, 128, 416), None),
    ("pose up T4x4s2 128->64 ->64x208", [128], 64, 4, 2, 1, 0, True, 1, (4, 32, 104), None),
    ("iconv7 3x3s1 1024->512 @4x13", [512, 512], 512, 3, 1, 1, 0, False, 1, (4, 4, 13), None),
    ("iconv6 3x3s1 1024->512 @8x26", [512, 512], 512, 3, 1, 1, 0, False, 1, (4, 8, 26), None),
    ("iconv5 3x3s1 512->256 @16x52", [256, 256], 256, 3, 1, 1, 0, False, 1, (4, 16, 52), None),
    ("iconv3 3x3s1 129->64 @64x208", [64, 64, 1], 64, 3, 1, 1, 0, False, 1, (4, 64, 208), None),
    ("conv7.2 3x3s1 512->512 @2x7", [512], 512, 3, 1, 1, 0, False, 1, (4, 2, 7), None),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
iters = int(os.environ.get("CB_ITERS", "10"))
for name, segs, cout, k, s, p, op, tr, act, (n, h, w), ohw in LAYERS:
    if flt not in name:
        continue
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, device="cuda", requires_grad=True) for c in segs]
    wt = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    cfg = (k, s, p, op, tr, act, 1.0, 0.0, ohw)
    out = ConvFn.apply(wt, b, cfg, *xs)
    g = torch.randn_like(out)
    out.backward(g)
    torch.cuda.synchronize()
    L.TIMER = L.KernelTimer()
    for _ in range(iters):
        out = ConvFn.apply(wt, b, cfg, *xs)
        out.backward(g)
    summ = L.TIMER.summary()
    L.TIMER = None
    line = f"{name:36s}"
    for kind in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        d = summ[kind]
        ms = d["ms"] / d["calls"]
        line += f" | {kind[5:]:5s} {ms*1e3:7.1f} us {d['flops']/d['calls']/(ms*1e-3)/1e12:6.1f} TF"
    print(line, flush=True)
