import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import se3_generate
g = np.load(os.path.join(ROOT, "tests/golden/se3_expmap.npz"))
vec = torch.tensor(g["vec"]).cuda().view(-1, 6, 1, 1).requires_grad_(True)
out = se3_generate.generate_se3(vec)
print("out err per row", (out.cpu().detach().view(6, 16) - torch.tensor(g["out"]).view(6, 16)).abs().max(1).values)
(out * torch.tensor(g["wt"]).cuda().float()).sum().backward()
print(vec.grad.view(6, 6).cpu()); print(torch.tensor(g["g_vec"]))
