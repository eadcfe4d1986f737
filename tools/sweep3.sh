for L in conv2.2 conv3.2 iconv4 conv4.2 "pose up"; do
for P in "0,0,0,0,0,0,0" "2,2,1,0,0,0,0" "2,2,1,0,1,0,0"; do
echo "== $L plan $P"; DVF_PIPE_DEBUG=1 DVF_PIPE_PLAN=$P CB_ITERS=5 timeout -k 10 120 python tools/conv_bench.py "$L" 2>&1 | grep -v "amdgpu.ids" | awk '/^\[pipe\]/{if(!s[$0]++)print substr($0,40,170)} !/^\[pipe\]/{print substr($0,1,100)}'
done; done
