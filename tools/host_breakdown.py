"""Host time of one training step by piece (wall-clock accumulators around the Python helpers; the GPU runs async).
Answers: of the 6 ms/step the host spends enqueueing, how much is ctypes calls, stream bookkeeping, allocation, glue?"""
import os, sys, time, argparse, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd")]
import torch, bench
from dvf import lib as L, conv as C, engine as E
args = argparse.Namespace(batch=4, height=256, width=832, seed=0, no_graph=True, force_ddp=False, graph_ddp=False)
step, fwd_bwd, opt, ddp = bench.build(args, bench.CONFIGS[2], torch.device("cuda", 0), 1, 0)
for _ in range(5): step()
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(obj, name, key=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[key or name]; e[0] += time.perf_counter() - t; e[1] += 1
    setattr(obj, name, g)
lib = L.lib()
class LibProxy:
    def __getattr__(self, n):
        f = getattr(lib, n)
        def g(*a):
            t = time.perf_counter(); r = f(*a); e = acc["ctypes:" + n]; e[0] += time.perf_counter() - t; e[1] += 1; return r
        return g
proxy = LibProxy()
L.lib = lambda: proxy
for n in ("dev", "ptr_array", "int_array", "stream", "check", "timed", "note_plans"):
    wrap(L, n, "lib." + n)
wrap(C, "_packed_weights"); wrap(E.FlatAdam, "fork_wgrad"); wrap(E.FlatAdam, "grad_ready"); wrap(E.FlatAdam, "step", "FlatAdam.step")
wrap(E.FlatAdam, "zero_grad"); wrap(torch, "empty_like", "torch.empty_like"); wrap(torch, "empty", "torch.empty")
fb = C.ConvFn.backward
def timed_bwd(ctx, g):
    t = time.perf_counter(); r = fb(ctx, g); e = acc["ConvFn.backward (total)"]; e[0] += time.perf_counter() - t; e[1] += 1; return r
C.ConvFn.backward = staticmethod(timed_bwd)
ff = C.ConvFn.forward
def timed_fwd(ctx, *a):
    t = time.perf_counter(); r = ff(ctx, *a); e = acc["ConvFn.forward (total)"]; e[0] += time.perf_counter() - t; e[1] += 1; return r
C.ConvFn.forward = staticmethod(timed_fwd)
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("enqueue %.2f ms/step (instrumented)" % (1e3 * (t1 - t0) / K))
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-40s %7.3f ms/step  %6.1f calls/step  %6.2f us/call" % (k, 1e3 * t / K, n / K, 1e6 * t / max(n, 1)))
