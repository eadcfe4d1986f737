"""Normalised per-family table from the three PMC passes of tools/pmc_step.sh (gpurun_out/pmc_step/p*/**/counter_collection.csv).
Kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs); MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
kernel cycles); LDS array busy per CU = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); conflict share = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE; wave-issue shares of SQ_WAVE_CYCLES.  usage: pmc_table.py <dir> [steps in the trace = 3]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(conv_pipe_kernel|wgrad_pipe_kernel|conv_gather_kernel|head_fwd_kernel|head_dgrad_kernel|head_wgrad_kernel|dconvt_s2_fwd_kernel|photo_fwd_kernel|photo_bwd_kernel|splitk_reduce)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
print("# tools/pmc_step.sh + tools/pmc_table.py: three rocprofv3 --pmc passes (+ --kernel-trace only) over `bench.py --steps 2 --warmup 1")
print("# --no-graph --serialize ...` (cfg 2, %d steps in the trace), summed per kernel family; normalisation in tools/pmc_table.py" % steps)
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    if cyc <= 0: continue
    wc = max(c["SQ_WAVE_CYCLES"], 1)
    print("%-22s kernel Mcycles/step %6.2f | MFMA pipe busy %.3f | LDS array busy/CU %.3f (conflict share %.2f) | wave issue: active %.3f wait_inst %.3f wait_any %.3f | insts/step: mfma %.3g valu %.3g lds %.3g" % (
        k, cyc / steps / 1e6, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc),
        c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc,
        c["SQ_INSTS_MFMA"] / steps, c["SQ_INSTS_VALU"] / steps, c["SQ_INSTS_LDS"] / steps))
