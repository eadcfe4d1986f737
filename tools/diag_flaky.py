"""Diagnostic (GPU box): run the golden cfg-2 step test body several times, print the post-Adam norm of conv1.0.bias."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from oracle import nets as onets
import DispNetS, PoseExpNet
from dvf.engine import FlatAdam
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
from dvf import lib as L
g = load_golden("step_unsup")
keys = [str(k) for k in g["p_disp_keys"]]
ref = dict(zip(keys, g["p_disp_norms"]))
for trial in range(6):
    L.SERIALIZE = trial >= 4
    sync = trial in (2, 3)
    disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
    disp.load_state_dict(onets.fill_params(onets.dispnet_layers(), seed=1)); pose.load_state_dict(onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2))
    disp.cuda().train(); pose.cuda().train()
    opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
    batch = synthetic_batch(2, 64, 128, seed=1234, device="cuda")
    for it in range(2):
        loss, terms = unsupervise_losses(disp, pose, batch)
        opt.zero_grad(); loss.backward()
        if sync: torch.cuda.synchronize()
        opt.step()
        if sync: torch.cuda.synchronize()
    sd = {k: v.detach().double().cpu() for k, v in disp.state_dict().items()}
    worst = max((abs(float(sd[k].norm()) - ref[k]) / ref[k], k) for k in keys)
    print(f"trial {trial} serialize={L.SERIALIZE} sync={sync}: worst norm relerr {worst[0]:.2e} at {worst[1]}; conv1.0.bias {abs(float(sd['conv1.0.bias'].norm())-ref['conv1.0.bias'])/ref['conv1.0.bias']:.2e}", flush=True)
