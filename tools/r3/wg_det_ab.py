"""wgrad per layer: deterministic (scratch + ordered reduce) vs atomic flush; HIP events around 20 back-to-back calls"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
import torch
from dvf import lib as L
LAYERS = [("conv1.2 7x7 32->32 @128x416", [32], 32, 7, 1, 3, (4, 128, 416)), ("conv2.2 5x5 64->64 @64x208", [64], 64, 5, 1, 2, (4, 64, 208)),
          ("conv3.2 128->128 @32x104", [128], 128, 3, 1, 1, (4, 32, 104)), ("conv4.2 256->256 @16x52", [256], 256, 3, 1, 1, (4, 16, 52)),
          ("conv5.2 512->512 @8x26", [512], 512, 3, 1, 1, (4, 8, 26)), ("conv6.2 512->512 @4x13", [512], 512, 3, 1, 1, (4, 4, 13)),
          ("iconv3 129->64 @64x208", [64, 64, 1], 64, 3, 1, 1, (4, 64, 208)), ("iconv2 65->32 @128x416", [32, 32, 1], 32, 3, 1, 1, (4, 128, 416)),
          ("conv3.0 s2 64->128", [64], 128, 3, 2, 1, (4, 64, 208))]
lib = L.lib()
for name, segs, cout, k, s, p, (n, h, w) in LAYERS:
    cin = sum(segs); oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    d = L.ConvDesc(n, cin, h, w, cout, oh, ow, k, k, s, p, 0, 1, 1.0, 0.0)
    xs = [torch.randn(n, c, h, w, device="cuda") for c in segs]
    dpre = torch.randn(n, cout, oh, ow, device="cuda")
    dw = torch.zeros(cout, cin, k, k, device="cuda"); db = torch.zeros(cout, device="cuda")
    wsf = int(lib.dvf_conv2d_wgrad_ws_floats(ctypes.byref(d), L.int_array(segs), len(segs)))
    ws = torch.empty(max(wsf, 1), device="cuda")
    res = {}
    for mode in ("atomic", "det"):
        def call():
            L.check(lib.dvf_conv2d_wgrad_det(ctypes.byref(d), L.ptr_array(xs), L.int_array(segs), len(segs), L.dev(dpre), L.dev(dw), 1, L.dev(db), 1,
                                             L.dev(ws) if mode == "det" else None, wsf if mode == "det" else 0, L.stream()), "wgrad")
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:30s} atomic {res['atomic']:7.1f} us | deterministic {res['det']:7.1f} us | scratch {wsf * 4 / 1e6:6.1f} MB", flush=True)
