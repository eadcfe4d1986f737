# Round-end check on the GPU box: micro-benchmark of LDS accumulate, whole -m gpu suite in one process, smoke, default bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic.hip -o /tmp/lds_atomic && /tmp/lds_atomic > $O/lds_atomic.txt 2>&1 &&
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1 ; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 && tail -2 $O/smoke.txt &&
python bench.py > $O/bench_default.json 2> $O/bench_default.err && cat $O/bench_default.json
