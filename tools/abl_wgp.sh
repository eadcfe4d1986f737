# rocprofv3 kernel durations of wgrad_pipe_kernel with DVF_WG_DBG ablations (tuning build): 0 full, 1 no DMA loads, 4 no MFMA,
# 8 no atomic flush, 5 = 1+4 (barriers + prologue + flush only), 13 = nothing but the item loop
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so
for L in $LAYERS; do for D in $DBGS; do
  rm -rf /tmp/kd; DVF_WG_DBG=$D timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -- python3 $R/tools/prof_one.py "$L" wgrad > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kd/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "wgrad" in n:
            d[n.split("(")[0][-40:] + " grid=%s lds=%s" % (r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size","?"), r.get("LDS_Block_Size","?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$L DBG=$D", {k: (len(v), round(sorted(v)[len(v) // 2], 1)) for k, v in d.items()})
PY
done; done
