"""Diagnostic (GPU box): where does the full-size cfg-2 step differ from the oracle?  Prints per-parameter
|g - ref32|, |g - ref64|, |ref32 - ref64| (relative to |ref32|) and the same for the gradient at the disparity output."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT)
from oracle import nets as onets, steps as osteps, losses as ol
import DispNetS, PoseExpNet
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
from dvf import lib as L
B, H, W = int(os.environ.get("B", 4)), int(os.environ.get("H", 256)), int(os.environ.get("W", 832))
if os.environ.get("SER"): L.SERIALIZE = True
from dvf import conv as _C
if os.environ.get("FUSE") == "0": _C.FUSE_RELU_BWD = False
if os.environ.get("DET") == "1": _C.set_deterministic(True)
dsd = onets.fill_params(onets.dispnet_layers(), seed=int(os.environ.get("WSEED", 1)))
psd = onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2)
disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
disp.load_state_dict({k: v.clone() for k, v in dsd.items()}); pose.load_state_dict({k: v.clone() for k, v in psd.items()})
disp.cuda().train(); pose.cuda().train()
batch = synthetic_batch(B, H, W, seed=int(os.environ.get("SEED", 1234)), device="cuda")
# --- HIP, with the disparity output's gradient retained
import loss_functions as LF
from dvf.conv import reciprocal
R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
_, T21 = pose((R2, R1))
d0 = disp(R2)[0]; d0.retain_grad(); T21.retain_grad()
depth = reciprocal(d0, 1e-4).squeeze(1)
il = LF.photometric_reconstruction_loss(R2, R1, L2, depth, T21, batch["T_R2L"], batch["K"], batch["Kinv"], img_scale=0.004) if not os.environ.get("PRESCALE") else LF.photometric_reconstruction_loss(0.004 * R2, 0.004 * R1, 0.004 * L2, depth, T21, batch["T_R2L"], batch["K"], batch["Kinv"])
sm = LF.smooth_loss(depth.unsqueeze(1))
(il + 10 * sm).backward()
torch.cuda.synchronize()
hip = {"disp": {k: p.grad.double().cpu() for k, p in disp.named_parameters() if p.grad is not None},
       "pose": {k: p.grad.double().cpu() for k, p in pose.named_parameters() if p.grad is not None}}
hip_d0, hip_T = d0.grad.double().cpu(), T21.grad.double().cpu()
hip_out = d0.detach().double().cpu()

def oracle(dt):
    ds = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in dsd.items()}
    ps = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in psd.items()}
    bt = osteps.synthetic_batch(B, H, W, seed=int(os.environ.get("SEED", 1234)), dtype=dt)
    r2, r1, l2 = bt["img_R2"], bt["img_R1"], bt["img_L2"]
    o0 = onets.dispnet_forward(ds, r2)[0]; o0.retain_grad()
    _, t21 = onets.posenet_forward(ps, torch.cat((r2, r1), 1), 2, True, sfm=False); t21.retain_grad()
    dep = (1 / (o0 + 1e-4)).squeeze(1)
    a = ol.photometric_reconstruction_loss(0.004 * r2, 0.004 * r1, 0.004 * l2, dep, t21, bt["T_R2L"], bt["K"], bt["Kinv"])
    s = ol.smooth_loss(dep.unsqueeze(1))
    (a + 10 * s).backward()
    return ({"disp": {k: v.grad.double() for k, v in ds.items() if v.grad is not None},
             "pose": {k: v.grad.double() for k, v in ps.items() if v.grad is not None}}, o0.grad.double(), t21.grad.double(), o0.detach().double())

cache = os.environ.get("ORACLE_CACHE")      # several HIP configurations against one oracle run (same B, H, W)
if cache and os.path.exists(cache):
    (r32, d32, t32, o32), (r64, d64, t64, o64) = torch.load(cache)
else:
    r32, d32, t32, o32 = oracle(torch.float32)
    r64, d64, t64, o64 = oracle(torch.float64)
    if cache: torch.save(((r32, d32, t32, o32), (r64, d64, t64, o64)), cache)
def rel(a, b, den): return float((a - b).norm() / den.norm())
print("disp out : hip-32 %.2e hip-64 %.2e 32-64 %.2e" % (rel(hip_out, o32, o32), rel(hip_out, o64, o32), rel(o32, o64, o32)))
print("g_disp0  : hip-32 %.2e hip-64 %.2e 32-64 %.2e" % (rel(hip_d0, d32, d32), rel(hip_d0, d64, d32), rel(d32, d64, d32)))
print("g_T21    : hip-32 %.2e hip-64 %.2e 32-64 %.2e" % (rel(hip_T, t32, t32), rel(hip_T, t64, t32), rel(t32, t64, t32)))
for n in ("disp", "pose"):
    for k in hip[n]:
        print("%-5s %-24s hip-32 %.2e hip-64 %.2e 32-64 %.2e" % (n, k, rel(hip[n][k], r32[n][k], r32[n][k]), rel(hip[n][k], r64[n][k], r32[n][k]), rel(r32[n][k], r64[n][k], r32[n][k])))

if os.environ.get("SUMMARY"):
    keys = ["iconv1.0", "upconv2.0", "iconv3.0", "upconv3.0", "predict_disp4.0", "iconv4.0", "upconv4.0", "iconv5.0", "iconv6.0", "iconv7.0", "conv7.0", "conv6.0", "conv5.0", "conv4.0", "conv3.0"]
    print("SUMMARY " + os.environ["SUMMARY"] + ": " + "  ".join("%s %.2f" % (k, rel(hip["disp"][k + ".weight"], r64["disp"][k + ".weight"], r32["disp"][k + ".weight"]) / rel(r32["disp"][k + ".weight"], r64["disp"][k + ".weight"], r32["disp"][k + ".weight"])) for k in keys))
