"""cfg-1 step (DispNetS depth-only, 128x416, batch 1): after ONE Adam step, which parameters moved the other way than in the
CPU oracle's run (|dev| > lr), and how far is each gradient from the oracle's?  (Diagnosis of the second-iteration loss.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nets as onets, steps as osteps
import DispNetS
from dvf.engine import FlatAdam
from dvf.steps import depth_only_losses
from dvf.synthetic import synthetic_batch
b, h, w = 1, 128, 416
batch = synthetic_batch(b, h, w, seed=1234, device="cuda")
cpu_batch = osteps.synthetic_batch(b, h, w, seed=1234)
dsd = onets.fill_params(onets.dispnet_layers(), seed=1)
disp = DispNetS.DispNetS(); disp.load_state_dict({k: v.clone() for k, v in dsd.items()}); disp.cuda().train()
opt = FlatAdam(list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
for it in range(2):
    loss, terms = depth_only_losses(disp, batch)
    opt.zero_grad(); loss.backward(); opt.join_wgrad()
    ref, grads, st = osteps.step_depth_only(dsd, cpu_batch, st if it else None)
    print("it", it, {k: (float(terms[k]), float(ref[k])) for k in ("img", "smooth", "total")})
    if it == 0:
        for k, p in disp.named_parameters():
            g, r = p.grad.detach().cpu().double(), grads["disp"][k].double()
            e = float((g - r).norm() / r.norm().clamp_min(1e-30))
            if e > 2e-5: print("  grad %-24s rel err %.2e  |r| %.3e" % (k, e, float(r.norm())))
    opt.step()
    if it == 0:
        tot = 0
        for k, v in disp.state_dict().items():
            dev = (v.detach().cpu().double() - dsd[k].double()).abs()
            n = int((dev > 1e-3).sum())
            tot += n
            if n: print("  flipped %-24s %7d of %8d  max dev %.2e" % (k, n, dev.numel(), float(dev.max())))
        print("  total flipped", tot)
