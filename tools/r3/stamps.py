"""In-kernel cycle account of conv_pipe_kernel per layer (tuning build, DVF_STAMPS): where the MFMA waves and the DMA
producers of a block spend their cycles.  usage: stamps.py [layer filter]   (env DVF_PIPE_X4, DVF_PIPE_PLAN as usual)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DVF_LIB", os.path.join(ROOT, "depth-vo-feat_amd/dvf/libdvf_hip_tuning.so"))
KIND = os.environ.get("STAMP_KIND", "pipe")      # pipe: conv_pipe_kernel (fwd / dgrad); wgrad: wgrad_pipe_kernel
os.environ["DVF_WG_STAMPS" if KIND == "wgrad" else "DVF_STAMPS"] = "1"
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from dvf.conv import ConvFn
from dvf import lib as L

LAYERS = [  # name, cin segs, cout, k, s, p, op, transposed, act, (N,H,W)
    ("conv1.2 7x7s1 32->32 @128x416", [32], 32, 7, 1, 3, 0, False, 1, (4, 128, 416)),
    ("conv2.0 5x5s2 32->64", [32], 64, 5, 2, 2, 0, False, 1, (4, 128, 416)),
    ("conv2.2 5x5s1 64->64 @64x208", [64], 64, 5, 1, 2, 0, False, 1, (4, 64, 208)),
    ("conv3.0 3x3s2 64->128", [64], 128, 3, 2, 1, 0, False, 1, (4, 64, 208)),
    ("conv3.2 3x3s1 128->128 @32x104", [128], 128, 3, 1, 1, 0, False, 1, (4, 32, 104)),
    ("conv4.2 3x3s1 256->256 @16x52", [256], 256, 3, 1, 1, 0, False, 1, (4, 16, 52)),
    ("conv5.2 3x3s1 512->512 @8x26", [512], 512, 3, 1, 1, 0, False, 1, (4, 8, 26)),
    ("conv6.2 3x3s1 512->512 @4x13", [512], 512, 3, 1, 1, 0, False, 1, (4, 4, 13)),
    ("iconv4 3x3s1 256->128 @32x104", [128, 128], 128, 3, 1, 1, 0, False, 1, (4, 32, 104)),
    ("iconv3 3x3s1 129->64 @64x208", [64, 64, 1], 64, 3, 1, 1, 0, False, 1, (4, 64, 208)),
    ("iconv2 3x3s1 65->32 @128x416", [32, 32, 1], 32, 3, 1, 1, 0, False, 1, (4, 128, 416)),
    ("upconv4 T3x3s2 256->128 ->32x104", [256], 128, 3, 2, 1, 1, True, 1, (4, 16, 52)),
    ("upconv3 T3x3s2 128->64 ->64x208", [128], 64, 3, 2, 1, 1, True, 1, (4, 32, 104)),
    ("upconv2 T3x3s2 64->32 ->128x416", [64], 32, 3, 2, 1, 1, True, 1, (4, 64, 208)),
    ("pose up T4x4s2 128->64 ->64x208", [128], 64, 4, 2, 1, 0, True, 1, (4, 32, 104)),
]
lib = L.lib()
lib.dvf_tuning_read_stamps.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * (8 * 16384))()


def account(tag):
    nb = lib.dvf_tuning_read_stamps(buf, 16384)
    if nb <= 0:
        print(f"  {tag:6s} (no pipelined launch)")
        return
    s = np.frombuffer(buf, dtype=np.uint64, count=8 * nb).reshape(nb, 8).astype(np.float64)
    s = s[s[:, 4] > 0]                      # blocks that left early (tile outside the class) wrote nothing
    if len(s) == 0:
        print(f"  {tag:6s} (no stamps)")
        return
    med = np.median(s, axis=0)
    tot = med[0] + med[1] + med[3]
    clock = tot / (med[4] * 10.0)          # cycles per ns
    us = lambda c: c / clock / 1e3
    pb = (ctypes.c_int * 96)()
    npl = lib.dvf_conv2d_last_plans(pb, 96)
    plans = " ".join("[k%d MT%d NT%d WM%d CK%d TBU%d KS%d BN%d NST%d thr%d lds%d m%x]" % tuple(pb[i:i + 12]) for i in range(0, npl, 12) if pb[i] == 1)
    if KIND == "wgrad":
        plans = " ".join("[k%d MT|TILE %x NTW%d x4 %d CK%d S%d NPIq%d blocks%d lds%d T%d RSq%d nseg%d]" % tuple(pb[i:i + 12]) for i in range(0, npl, 12) if pb[i] == 8)
        tot = med[0] + med[1]
        clock = tot / (med[4] * 10.0)
        us = lambda c: c / clock / 1e3
        print(f"  {tag:6s} blocks {len(s):4d} | block {us(tot):6.1f} us @ {clock:4.2f} GHz = prologue {us(med[0]):5.1f} + loop {us(med[1]):6.1f} "
              f"(MFMA waves at barriers {us(med[2]):5.1f}, flushing {us(med[3]):5.1f}) | producer: DMA wait {us(med[5]):5.1f}, at barriers {us(med[6]):6.1f}, issuing {us(med[7]):5.1f} {plans}")
        return
    print(f"  {tag:6s} blocks {len(s):4d} | block {us(tot):6.1f} us @ {clock:4.2f} GHz = prologue {us(med[0]):5.1f} + loop {us(med[1]):6.1f} "
          f"(MFMA waves at barriers {us(med[2]):5.1f}) + epilogue {us(med[3]):5.1f} | producer: DMA wait {us(med[5]):5.1f}, at barriers {us(med[6]):6.1f}, issuing {us(med[7]):5.1f} {plans}")


flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, segs, cout, k, s, p, op, tr, act, (n, h, w) in LAYERS:
    if flt not in name:
        continue
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, device="cuda", requires_grad=True) for c in segs]
    wt = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    cfg = (k, s, p, op, tr, act, 1.0, 0.0, None)
    print(name, flush=True)
    for rep in range(2):                   # second pass: warm
        out = ConvFn.apply(wt, b, cfg, *xs)
        torch.cuda.synchronize()
        if rep and KIND == "pipe": account("fwd")
        g = torch.randn_like(out)
        out.backward(g)
        torch.cuda.synchronize()
        if rep: account("dgrad" if KIND == "pipe" else "wgrad")
