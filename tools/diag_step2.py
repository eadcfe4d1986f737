"""Diagnostic (GPU box): two Adam steps of the cfg-2 body at 64x128, HIP vs CPU oracle, elementwise on small tensors."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT)
from oracle import nets as onets, steps as osteps
import DispNetS, PoseExpNet
from dvf.engine import FlatAdam
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
from dvf import lib as L
if os.environ.get("SER"): L.SERIALIZE = True
b, h, w = 2, 64, 128
dsd = onets.fill_params(onets.dispnet_layers(), seed=1); psd = onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2)
disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
disp.load_state_dict({k: v.clone() for k, v in dsd.items()}); pose.load_state_dict({k: v.clone() for k, v in psd.items()})
disp.cuda().train(); pose.cuda().train()
opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
batch = synthetic_batch(b, h, w, seed=1234, device="cuda")
obatch = osteps.synthetic_batch(b, h, w, seed=1234)
st = None
for it in range(2):
    loss, terms = unsupervise_losses(disp, pose, batch)
    opt.zero_grad(); loss.backward(); opt.join_wgrad(); L.join_aux_streams(); torch.cuda.synchronize()
    g_hip = {k: p.grad.detach().cpu().clone() for k, p in disp.named_parameters() if p.grad is not None}
    opt.step(); torch.cuda.synchronize()
    out, grads, st = osteps.step_unsupervise(dsd, psd, obatch, st)
    for k in ("conv1.0.bias", "conv7.2.bias", "predict_disp1.0.bias"):
        gh, go = g_hip[k], grads["disp"][k]
        ph, po = dict(disp.named_parameters())[k].detach().cpu(), dsd[k]
        i = int((ph - po).abs().argmax())
        print(f"it{it} {k}: grad relerr {float((gh-go).norm()/go.norm()):.2e}; param max|d| {float((ph-po).abs().max()):.3e} at {i}: g_hip {float(gh.flatten()[i]):.4e} g_ref {float(go.flatten()[i]):.4e} |g|max {float(go.abs().max()):.3e}")
