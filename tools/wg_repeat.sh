cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so
for D in $DBGS; do
rm -rf /tmp/kd; DVF_WG_DBG=$D timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -- python3 $R/tools/wg_repeat.py $ARGS > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/kd/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
print("DBG=$D", [round(d, 1) for _, d in rows], "gaps", [round((rows[i+1][0]-rows[i][0])/1e3 - rows[i][1], 1) for i in range(len(rows)-1)])
PY
done
