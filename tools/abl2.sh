cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in $LAYERS; do
for d in $DBGS; do
  DVF_DBG=$d timeout -k 10 100 rocprofv3 --kernel-trace -d $R/gpurun_out/abl2_${f}_$d -o r -- python3 $R/tools/prof_one.py $f fwd > /dev/null 2>&1
  echo "== $f DBG=$d"; python3 $R/tools/kstats.py $R/gpurun_out/abl2_${f}_$d/r_results.db | grep "conv_pipe\|gather\|reduce"
done; done
