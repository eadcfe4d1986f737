"""Per-layer kernel durations from rocprofv3 (not from event brackets): joins the call log of bench.py's recorded pass
(DVF_CALL_LOG: calls in launch order with the kernel families each launched) with the kernel trace of the same process.

    DVF_CALL_LOG=gpurun_out/x/calls.jsonl rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x/trace -o t -- \
        python3 bench.py --steps 2 --warmup 1 --no-graph --serialize --no-cpu-baseline
    python3 tools/r3/launch_table.py gpurun_out/x/calls.jsonl gpurun_out/x/trace/t_kernel_trace.csv

The recorded pass is the LAST eager step of the process, so the last n dispatches of each kernel family are its launches.
A call's time = its family's kernels + the split-K reduce launches issued between them and the next call's first kernel."""
import csv
import json
import re
import sys

FAMILY_OF = [("conv_pipe_kernel", "pipe"), ("wgrad_pipe_kernel", "wgrad_pipe"), ("conv_gather_kernel", "gather"),
             ("head_fwd_kernel", "head"), ("head_dgrad_kernel", "head"), ("head_wgrad_kernel", "head"),
             ("head_seg_dgrad", "head"), ("dconvt_s2_fwd_kernel", "dconvt"), ("conv_wgrad_kernel", "wgrad")]
HELPERS = ("splitk_reduce", "wgrad_reduce", "bias_finish", "head_wgrad_finish", "act_bwd", "bias_act")


def family(name):
    for pat, fam in FAMILY_OF:
        if pat in name:
            return fam
    return None


def main(calls_path, trace_path, peak=157.3):
    calls = [json.loads(l) for l in open(calls_path)]
    rows = sorted(csv.DictReader(open(trace_path)), key=lambda r: int(r["Start_Timestamp"]))
    disp = []
    for r in rows:
        n = r["Kernel_Name"]
        disp.append((family(n), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                     "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"], r["Grid_Size_Z"])))
    need = sum(len([k for k in c["kernels"]]) for c in calls if c["kind"].startswith("conv_"))
    # walk backwards: find the start of the recorded pass = the position where the remaining family dispatches == need
    fam_idx = [i for i, d in enumerate(disp) if d[0] is not None]
    if len(fam_idx) < need:
        sys.exit("trace holds fewer convolution kernels than the call log")
    pos = fam_idx[len(fam_idx) - need]
    out = []
    for c in calls:
        if not c["kind"].startswith("conv_"):
            continue
        t, names, grids = 0.0, [], []
        for fam in c["kernels"]:
            while disp[pos][0] is None:             # helper kernels before this launch belong to the previous call
                if out and any(h in disp[pos][1] for h in HELPERS):
                    out[-1]["helper_us"] += disp[pos][2]
                pos += 1
            if disp[pos][0] != fam:
                sys.exit(f"order mismatch at {c['tag']}: log says {fam}, trace has {disp[pos][1][:60]}")
            t += disp[pos][2]
            names.append(re.sub(r"void |dvfp::|\(anonymous namespace\)::|\(.*", "", disp[pos][1]))
            grids.append(disp[pos][3])
            pos += 1
        out.append({"kind": c["kind"], "tag": c["tag"], "family": c["family"], "flops": c["flops"], "kernel_us": t, "helper_us": 0.0,
                    "event_us": 1e3 * c["event_ms"], "names": names, "grids": grids})
    while pos < len(disp) and disp[pos][0] is None:
        if out and any(h in disp[pos][1] for h in HELPERS):
            out[-1]["helper_us"] += disp[pos][2]
        pos += 1
    print("%-11s %8s %8s %8s %7s %7s  %s" % ("kind", "kern us", "+help us", "event us", "TF/s", "frac", "layer | kernel | grid"))
    tot = {}
    for o in sorted(out, key=lambda o: -(o["kernel_us"] + o["helper_us"])):
        us = o["kernel_us"] + o["helper_us"]
        tf = o["flops"] / us / 1e6 if us > 0 else 0.0
        print("%-11s %8.1f %8.1f %8.1f %7.1f %7.3f  %s | %s | %s" % (o["kind"], o["kernel_us"], o["helper_us"], o["event_us"], tf, tf / peak,
                                                                  o["tag"], ",".join(o["names"]), ",".join(o["grids"])))
        d = tot.setdefault(o["kind"] + "/" + o["family"], [0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += us; d[2] += o["flops"]; d[3] += o["event_us"]
    print()
    for k, (n, us, fl, ev) in sorted(tot.items()):
        print("%-24s %3d calls  %8.1f us kernels (%8.1f us by events)  %6.1f TF/s  frac %.3f" % (k, n, us, ev, fl / us / 1e6, fl / us / 1e6 / peak))


if __name__ == "__main__":
    main(*sys.argv[1:3])
