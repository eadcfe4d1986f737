for f in "conv3.2" "conv4.2" "conv5.2" "iconv4" "conv2.2"; do
for d in 0 1 2 3 4; do echo "== $f DBG=$d"; DVF_DBG=$d CB_ITERS=5 timeout -k 10 120 python tools/conv_bench.py "$f" 2>/dev/null | cut -c1-100; done; done
