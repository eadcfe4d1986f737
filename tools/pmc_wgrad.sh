# PMC passes over conv_wgrad_kernel of one layer (tuning build), DVF_WG_DBG in $DBGS.  Run through gpurun.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so
L=${LAYER:-iconv4}
for d in $DBGS; do
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH"; do
  i=$((i+1))
  DVF_WG_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/pmcw_${d}_$i -- python3 $R/tools/prof_one.py $L wgrad > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0]); dur = []
for f in glob.glob("$R/gpurun_out/pmcw_${d}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wgrad" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob("$R/gpurun_out/pmcw_${d}_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wgrad" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("DBG=$d", "$L", "kernel us", round(sum(dur) / max(len(dur), 1) / 1e3, 1), {k: round(v[0] / max(v[1], 1)) for k, v in sorted(agg.items())})
PY
done
