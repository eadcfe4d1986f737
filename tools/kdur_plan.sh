# rocprofv3 kernel durations of one layer direction under DVF_PIPE_PLAN overrides (tuning build)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so
for L in $LAYERS; do for W in $WHAT; do for P in $PLANS; do
  rm -rf /tmp/kd; if [ "$P" = "default" ]; then unset DVF_PIPE_PLAN; else export DVF_PIPE_PLAN=$P; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -- python3 $R/tools/prof_one.py "$L" $W > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections, re
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kd/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"(conv_pipe_kernel<[^>]*>|conv_gather_kernel<[^>]*>|splitk_reduce\w*|head_\w+)", n)
        if m:
            d[m.group(1) + " g%dx%sx%s lds%s" % (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("LDS_Block_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$L $W plan=$P", {k: round(sorted(v)[len(v) // 2], 1) for k, v in d.items()})
PY
done; done; done
