"""Diagnostic (GPU box): gradients of the feature network in the cfg-3 step at 64x128: HIP vs fp32 oracle vs fp64 oracle."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT)
from oracle import nets as onets, steps as osteps
import DispNetS, PoseExpNet, feat_extractor
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
b, h, w = 2, 64, 128
dsd = onets.fill_params(onets.dispnet_layers(), seed=1); psd = onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2)
fsd = onets.fill_params(onets.featnet_layers(), seed=3)
disp, pose, feat = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True), feat_extractor.FeatExtractor()
for m, sd in ((disp, dsd), (pose, psd), (feat, fsd)):
    m.load_state_dict({k: v.clone() for k, v in sd.items()}); m.cuda().train()
batch = synthetic_batch(b, h, w, seed=1234, device="cuda")
loss, terms = unsupervise_losses(disp, pose, batch, feat_extractor=feat)
loss.backward(); torch.cuda.synchronize()
hip = {k: p.grad.double().cpu() for k, p in feat.named_parameters()}
def run(dt):
    c = lambda sd: {k: v.detach().clone().to(dt) for k, v in sd.items()}
    out, grads, _ = osteps.step_unsupervise(c(dsd), c(psd), osteps.synthetic_batch(b, h, w, seed=1234, dtype=dt), feat_sd=c(fsd), do_update=False)
    return {k: v.double() for k, v in grads["feat"].items()}, out
r32, o32 = run(torch.float32); r64, o64 = run(torch.float64)
print("feat loss hip %.8f ref32 %.8f ref64 %.8f" % (float(terms["feat"]), float(o32["feat"]), float(o64["feat"])))
for k in hip:
    n = r32[k].norm()
    mx = r32[k].abs().max()
    print("%-22s hip-32 %.2e hip-64 %.2e 32-64 %.2e | max elem err/maxabs: hip %.2e ref32 %.2e" % (
        k, float((hip[k] - r32[k]).norm() / n), float((hip[k] - r64[k]).norm() / n), float((r32[k] - r64[k]).norm() / n),
        float((hip[k] - r64[k]).abs().max() / mx), float((r32[k] - r64[k]).abs().max() / mx)))
