# in-step layer table + bench line of the tuning build under a set of DVF_* knobs:  knob.sh <tag> VAR=VAL ...
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/knobs; mkdir -p $O; cd $R
env DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so "$@" DVF_LAYER_TABLE=400 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$TAG.json 2> $O/layers_$TAG.txt
python3 - <<PY
import json; d=json.load(open("$O/bench_$TAG.json")); print("$TAG", "$*", "ms/step %.3f" % d["ms_per_step"], d["kernel_ms_per_step"])
PY
