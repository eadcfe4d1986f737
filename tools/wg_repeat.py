"""Launch the wgrad of ONE single-segment layer REP times back to back (no other kernel in between) -- for rocprofv3
--kernel-trace: do repeated launches of the same kernel get cheaper (instruction cache warm)?
usage: wg_repeat.py cin cout k stride N H W [rep]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
from dvf import lib as L
cin, cout, k, s, n, h, w = map(int, sys.argv[1:8])
rep = int(sys.argv[8]) if len(sys.argv) > 8 else 6
p = k // 2
oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
x = torch.randn(n, cin, h, w, device="cuda")
dpre = torch.randn(n, cout, oh, ow, device="cuda")
dw = torch.zeros(cout, cin, k, k, device="cuda")
desc = L.ConvDesc(n, cin, h, w, cout, oh, ow, k, k, s, p, 0, 0, 1.0, 0.0)
torch.cuda.synchronize()
for _ in range(rep):
    L.check(L.lib().dvf_conv2d_wgrad(ctypes.byref(desc), L.ptr_array([x]), L.int_array([cin]), 1, L.dev(dpre), L.dev(dw), 1, L.stream()), "wgrad")
torch.cuda.synchronize()
print("ok")
