# Matrix-pipe and LDS counters per kernel family over the serialised cfg-2 step (two separate PMC passes; no trace domains
# besides --kernel-trace).  Output: gpurun_out/pmc_step.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_step; rm -rf $O; mkdir -p $O
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob, collections, os, re
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_step"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float)
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(conv_pipe_kernel|wgrad_pipe_kernel|conv_gather_kernel|head_fwd_kernel|dconvt_s2_fwd_kernel|photo_fwd_kernel|photo_bwd_kernel)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = open(O + "/../pmc_step.txt", "w")
def w(s):
    print(s); out.write(s + "\n")
w("# per kernel family, summed over the launches of 3 serialised cfg-2 steps (rocprofv3 --pmc, three passes)")
w("# mfma_busy/busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES ; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
for k, c in agg.items():
    g = lambda n: c.get(n, 0.0)
    w("%-22s launches %4d  mfma_busy/busy %.3f  mfma_busy/(4*gui_active) %.3f  lds_conflict %.3f  wait_inst/wave_cycles %.3f  wait_any/wave_cycles %.3f  insts: mfma %.3g valu %.3g lds %.3g" % (
        k, cnt[k].get("SQ_BUSY_CYCLES", 0), g("SQ_VALU_MFMA_BUSY_CYCLES") / max(g("SQ_BUSY_CYCLES"), 1), g("SQ_VALU_MFMA_BUSY_CYCLES") / max(4 * g("GRBM_GUI_ACTIVE"), 1),
        g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
        g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS")))
PY
