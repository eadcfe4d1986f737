import os, sys
ROOT="/root/repo"; sys.path[:0]=[ROOT, os.path.join(ROOT,"depth-vo-feat_amd")]
import torch
from dvf.conv import ConvFn
torch.manual_seed(0)
for (n,c,h,w,co) in [(1,8,24,40,16),(2,64,16,32,64),(1,32,24,40,32)]:
    x=torch.randn(n,c,h,w,device="cuda"); wt=torch.randn(co,c,3,3,device="cuda",requires_grad=True); b=torch.zeros(co,device="cuda",requires_grad=True)
    out=ConvFn.apply(wt,b,(3,1,1,0,False,0,1.0,0.0,None),x)
    g=torch.randn_like(out); out.backward(g)
    ref=torch.nn.grad.conv2d_weight(x.cpu().double(), wt.shape, g.cpu().double(), padding=1)
    err=(wt.grad.cpu().double()-ref).abs().max()/ref.abs().max()
    print((n,c,h,w,co), float(err))
    # which (m, c) wrong
    d=(wt.grad.cpu().double()-ref).abs().amax(dim=(2,3))/ref.abs().max()
    bad=(d>1e-4).nonzero()
    print(" bad m:", sorted(set(bad[:,0].tolist()))[:20], " n bad", len(bad))
