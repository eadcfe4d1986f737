# kernel durations (rocprofv3 --kernel-trace) of one layer direction, warm (same tensors re-used) vs cold (caches flushed)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in $LAYERS; do for W in $WHAT; do for F in "" 1; do
  rm -rf /tmp/kd; FLUSH=$F timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -- python3 $R/tools/prof_one.py "$L" $W > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kd/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("conv_pipe", "conv_wgrad", "conv_gather", "splitk", "head_")):
            d[n.split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$L $W", "cold" if "$F" else "warm", {k: round(sorted(v)[len(v) // 2], 1) for k, v in d.items()})
PY
done; done; done
