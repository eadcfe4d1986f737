"""Diagnostic (GPU box): loss trajectory of the cfg-2 body over N Adam steps, HIP vs CPU oracle, at 64x128 batch 2."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT)
from oracle import nets as onets, steps as osteps
import DispNetS, PoseExpNet
from dvf.engine import FlatAdam
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
b, h, w = 2, 64, 128
torch.manual_seed(0)
disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
disp.init_weights(); pose.init_weights()
dsd = {k: v.detach().clone() for k, v in disp.state_dict().items()}
psd = {k: v.detach().clone() for k, v in pose.state_dict().items()}
disp.cuda().train(); pose.cuda().train()
opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
batch = synthetic_batch(b, h, w, seed=1234, device="cuda")
cb = osteps.synthetic_batch(b, h, w, seed=1234)
st = None
for it in range(n):
    loss, terms = unsupervise_losses(disp, pose, batch)
    opt.zero_grad(); loss.backward(); opt.step()
    ref, _, st = osteps.step_unsupervise(dsd, psd, cb, st)
    if it < 5 or it % 5 == 4:
        print(f"step {it:3d}  hip total {float(terms['total']):12.5f} img {float(terms['img']):.6f}   oracle total {float(ref['total']):12.5f} img {float(ref['img']):.6f}", flush=True)
