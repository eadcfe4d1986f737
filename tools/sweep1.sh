for L in conv5.2 conv4.2 conv3.2 iconv4; do
for P in "0,0,0,0,0,0,0" "0,0,0,16,0,0,2" "0,0,0,8,8,0,2" "0,0,0,8,2,0,3" "0,0,0,16,2,0,2" "0,0,0,8,1,0,3" "0,0,0,16,1,0,2"; do
echo "== $L plan $P"; DVF_PIPE_DEBUG=1 DVF_PIPE_PLAN=$P CB_ITERS=5 timeout -k 10 120 python tools/conv_bench.py "$L" 2>&1 | grep -v "amdgpu.ids" | awk '/^\[pipe\]/{if(!s[$0]++)print substr($0,1,200)} !/^\[pipe\]/{print substr($0,1,110)}'
done; done
