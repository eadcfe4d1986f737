for L in conv5.2 conv4.2 conv3.2 iconv4; do
for P in "0,0,0,0,0,0,0" "0,0,0,8,8,0,2" "0,0,0,8,4,0,2" "0,0,0,8,2,0,2" "0,0,0,8,2,0,3" "0,0,0,4,8,0,3"; do
echo "== $L plan $P"; DVF_PIPE_PLAN=$P CB_ITERS=5 timeout -k 10 120 python tools/conv_bench.py "$L" 2>&1 | grep -v "amdgpu.ids" | cut -c1-100
done; done
