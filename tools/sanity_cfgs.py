"""Full-size sanity run of the three step bodies (no parity, just finite losses + timing): cfg 2 / cfg 3 (feature loss) /
cfg 4 (train.py 4-scale sfm loss) at 256x832 and cfg 2 at 384x1280."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd")]
import torch
import DispNetS, PoseExpNet, PoseExpNet_sfm, feat_extractor
from dvf.engine import FlatAdam
from dvf.steps import unsupervise_losses, train_sfm_losses, unsupervise_dvo_losses
from dvf.synthetic import synthetic_batch
dev = "cuda"
def run(name, h, w, b, make):
    torch.manual_seed(0)
    nets, loss_fn = make()
    for n in nets:
        n.init_weights() if hasattr(n, "init_weights") else None
        n.to(dev).train()
    opt = FlatAdam([p for n in nets for p in n.parameters()], lr=1e-4)
    batch = synthetic_batch(b, h, w, seed=1, device=dev)
    for i in range(4):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, terms = loss_fn(batch)
        opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    ok = bool(torch.isfinite(loss)) and all(bool(torch.isfinite(p).all()) for n in nets for p in n.parameters())
    print(f"{name:34s} {h}x{w} b{b}: loss {float(loss):.4f} finite={ok}  {ms:.1f} ms/step (eager)", flush=True)
    assert ok
d = lambda: DispNetS.DispNetS()
run("cfg2 unsupervise", 256, 832, 4, lambda: ((lambda dn, pn: ([dn, pn], lambda bt: unsupervise_losses(dn, pn, bt)))(d(), PoseExpNet.PoseExpNet(output_exp=True))))
run("cfg3 unsupervise + feature", 256, 832, 4, lambda: ((lambda dn, pn, fe: ([dn, pn, fe], lambda bt: unsupervise_losses(dn, pn, bt, feat_extractor=fe)))(d(), PoseExpNet.PoseExpNet(output_exp=True), feat_extractor.FeatExtractor())))
run("cfg4 train_sfm (4 scales, masks)", 256, 832, 4, lambda: ((lambda dn, pn: ([dn, pn], lambda bt: train_sfm_losses(dn, pn, bt, w2=0.2)))(d(), PoseExpNet_sfm.PoseExpNet(nb_ref_imgs=2, output_exp=True))))
run("dvo (se3 + pixel warp)", 256, 832, 4, lambda: ((lambda dn, pn: ([dn, pn], lambda bt: unsupervise_dvo_losses(dn, pn, dict(bt, T_R2L=bt["T_R2L"][:, [3, 4, 5, 0, 1, 2]].contiguous()))))(d(), PoseExpNet.PoseExpNet(output_exp=True))))
run("cfg2 at 384x1280", 384, 1280, 2, lambda: ((lambda dn, pn: ([dn, pn], lambda bt: unsupervise_losses(dn, pn, bt)))(d(), PoseExpNet.PoseExpNet(output_exp=True))))
run("cfg2 odd size 192x640 b3", 192, 640, 3, lambda: ((lambda dn, pn: ([dn, pn], lambda bt: unsupervise_losses(dn, pn, bt)))(d(), PoseExpNet.PoseExpNet(output_exp=True))))
