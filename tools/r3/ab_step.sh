# A/B of two builds of the library inside one gpurun call: parity tests of the convolution kernels on the new build, then
# the in-step layer table + bench line of the new build and of the round-2 build (libdvf_hip_r2.so).
# usage: tools/r3/ab_step.sh <tag> [pytest args]
R=$GRAFT_REPO_ROOT; TAG=${1:-ab}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_conv_fuzz.py tests/test_gpu_bench_shapes.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.txt
for L in new r2; do
  if [ $L = r2 ]; then export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_r2.so; else unset DVF_LIB; fi
  DVF_LAYER_TABLE=400 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$L.json 2> $O/layers_$L.txt
  echo "$L: $(cut -c1-200 $O/bench_$L.json)"
done
