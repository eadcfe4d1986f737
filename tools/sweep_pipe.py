"""Plan sweep for conv_pipe_kernel (tuning build: DVF_LIB=.../libdvf_hip_tuning.so): for each layer of tools/conv_bench.py
matching the filter, time forward and dgrad under DVF_PIPE_PLAN=MT,NT,WM,CK,KS,BN,NST overrides and print the best plans
next to the planner's own choice.  usage: sweep_pipe.py [filter] ; env CB_ITERS"""
import itertools, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dvf.conv import ConvFn
from dvf import lib as L
assert L.lib().dvf_build_has_tuning(), "needs the tuning build (DVF_LIB)"
import importlib.util
spec = importlib.util.spec_from_file_location("cb", os.path.join(ROOT, "tools", "conv_bench.py"))
src = open(os.path.join(ROOT, "tools", "conv_bench.py")).read()
LAYERS = eval(src[src.index("LAYERS = ["):src.index("]\nflt")].replace("LAYERS = ", "") + "]")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
iters = int(os.environ.get("CB_ITERS", "6"))


def run(layer, plan):
    name, segs, cout, k, s, p, op, tr, act, (n, h, w), ohw = layer
    if plan is None:
        os.environ.pop("DVF_PIPE_PLAN", None)
    else:
        os.environ["DVF_PIPE_PLAN"] = ",".join(str(x) for x in plan)
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, device="cuda", requires_grad=True) for c in segs]
    wt = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    cfg = (k, s, p, op, tr, act, 1.0, 0.0, ohw)
    L.PLAN_LOG = set()
    try:
        out = ConvFn.apply(wt, b, cfg, *xs)
        g = torch.randn_like(out)
        out.backward(g)
        torch.cuda.synchronize()
    except RuntimeError:
        return None
    kinds = {(pl[0], pl[1]) for pl in L.PLAN_LOG}
    plans = sorted(pl for pl in L.PLAN_LOG if pl[0] in ("fwd", "dgrad"))
    L.PLAN_LOG = None
    if ("fwd", 1) not in kinds:          # forward did not run on conv_pipe_kernel under this plan
        return None
    L.TIMER = L.KernelTimer()
    for _ in range(iters):
        out = ConvFn.apply(wt, b, cfg, *xs)
        out.backward(g)
    summ = L.TIMER.summary()
    L.TIMER = None
    f = summ["conv_fwd"]["ms"] / summ["conv_fwd"]["calls"] * 1e3
    d = summ["conv_dgrad"]["ms"] / summ["conv_dgrad"]["calls"] * 1e3
    return f, d, plans


for layer in LAYERS:
    if flt not in layer[0]:
        continue
    base = run(layer, None)
    if base is None:
        print(f"{layer[0]:36s} not on conv_pipe_kernel"); continue
    print(f"{layer[0]:36s} planner: fwd {base[0]:7.1f} us dgrad {base[1]:7.1f} us   {[p[2:10] for p in base[2]]}", flush=True)
    res = []
    for MT, NT, WM, CK, KS, NST in itertools.product((1, 2), (1, 2), (1, 2), (4, 8, 16), (0, 1, 2, 4), (2, 3)):
        if MT == 2 and NT == 2:
            continue
        r = run(layer, (MT, NT, WM, CK, KS, 0, NST))
        if r is not None:
            res.append((r[0], r[1], (MT, NT, WM, CK, KS, NST)))
    for key, nm in ((0, "fwd"), (1, "dgrad")):
        best = sorted(res, key=lambda t: t[key])[:3]
        print("    best %-5s: " % nm + "  ".join(f"{t[key]:7.1f} us {t[2]}" for t in best), flush=True)
