# per-layer A/B of the patch DMA lane width (tuning build): DVF_PIPE_X4=0 (4-byte lanes) vs 1 (16-byte lanes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-x4}; mkdir -p $O; cd $R
export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so CB_ITERS=20
for X in 0 1; do
  DVF_PIPE_X4=$X timeout -k 10 300 python3 tools/conv_bench.py > $O/conv_bench_x4_$X.txt 2>&1
  echo "X4=$X rc $?"
done
paste -d'\n' $O/conv_bench_x4_0.txt $O/conv_bench_x4_1.txt | grep -v amdgpu.ids
