"""Build profiles/r01_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over
`bench.py --steps 3 --warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing`.
usage: traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <steps traced | auto> <out.json>
"auto" counts the optimizer steps in the trace itself (adam_prep_kernel runs once per step, warm-up steps included)."""
import collections, csv, glob, json, os, re, sys

GROUPS = ["conv_pipe", "conv_gather", "wgrad_pipe", "conv_wgrad", "conv_pack_batch", "splitk_reduce", "act_bwd", "bias_act", "photo_fwd",
          "photo_bwd", "smooth_fwd", "smooth_bwd", "adam", "fillBuffer"]


STEPS_SEEN = [0]


def collect(d):
    agg = collections.defaultdict(lambda: [0.0, 0])
    nprep = 0
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                     # gpurun merges every run's files into the same directory: newest only
        for r in csv.DictReader(open(f)):
            if "adam_prep" in r["Kernel_Name"]:
                nprep += 1
                continue
            for g in GROUPS:
                if g in r["Kernel_Name"]:
                    agg[g][0] += float(r["Counter_Value"])
                    agg[g][1] += 1
                    break
    STEPS_SEEN[0] = nprep
    return agg


fetch, write = collect(sys.argv[1]), collect(sys.argv[2])
steps = float(STEPS_SEEN[0]) if sys.argv[3] == "auto" else float(sys.argv[3])
print("steps in trace:", STEPS_SEEN[0], "-> normalising by", steps)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over `bench.py --steps 3 "
                 "--warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing` (cfg 2); values are per training step",
       "correction": "FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B "
                     "(MI355X_MICROARCH.md, HBM section; calibrated in round 1 on adam_kernel and act_bwd_kernel), so read "
                     "bytes = 2 x FETCH_SIZE x 1024 and write bytes = WRITE_SIZE x 1024.",
       "kernels": {}}
for g in GROUPS:
    if g in fetch:
        out["kernels"][g] = {"launches_per_step": fetch[g][1] / steps,
                             "read_bytes_per_step": 2 * 1024 * fetch[g][0] / steps,
                             "fetch_size_raw_bytes": 1024 * fetch[g][0] / steps,
                             "write_bytes_per_step": 1024 * write.get(g, [0, 0])[0] / steps}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: {a: round(b) for a, b in v.items()} for k, v in out["kernels"].items()}, indent=1))
