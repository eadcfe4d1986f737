import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("no cgroup cpu.max", e)
from oracle import nets as onets, steps as osteps
for nt in (16, 32):
    torch.set_num_threads(nt)
    dsd = onets.fill_params(onets.dispnet_layers(), seed=1)
    psd = onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2)
    batch = osteps.synthetic_batch(1, 256, 832, seed=1234)
    t0 = time.perf_counter(); _, _, st = osteps.step_unsupervise(dsd, psd, batch, None); t1 = time.perf_counter()
    _, _, st = osteps.step_unsupervise(dsd, psd, batch, st); t2 = time.perf_counter()
    print(nt, "threads: first step %.2fs second %.2fs" % (t1 - t0, t2 - t1), flush=True)
