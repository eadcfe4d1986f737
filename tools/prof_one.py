"""Run ONE conv layer direction repeatedly (for rocprofv3 --pmc).  usage: prof_one.py <layer filter> <fwd|dgrad|wgrad>"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("cb", os.path.join(ROOT, "tools", "conv_bench.py"))
src = open(os.path.join(ROOT, "tools", "conv_bench.py")).read()
ns = {"__file__": os.path.join(ROOT, "tools", "conv_bench.py")}
exec(src.split("flt = sys.argv")[0], ns)      # LAYERS table only
from dvf.conv import ConvFn
flt, what = sys.argv[1], sys.argv[2]
for name, segs, cout, k, s, p, op, tr, act, (n, h, w), ohw in ns["LAYERS"]:
    if flt not in name:
        continue
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, device="cuda", requires_grad=(what != "fwd")) for c in segs]
    wt = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5).requires_grad_(what == "wgrad")
    b = torch.zeros(cout, device="cuda")
    cfg = (k, s, p, op, tr, act, 1.0, 0.0, ohw)
    flush = torch.empty(160 << 20, device="cuda") if os.environ.get("FLUSH") else None     # 640 MB: beyond L2 + Infinity Cache
    for _ in range(5):
        if flush is not None:
            flush.zero_()
        out = ConvFn.apply(wt, b, cfg, *xs)
        if what != "fwd":
            out.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    print("done", name)
