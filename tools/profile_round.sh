# Profiles judged for the round (run on the GPU box through gpurun).  Outputs under gpurun_out/.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats -d $O/prof_serial -o serial -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > $O/bench_serial.json 2> $O/bench_serial.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
done
ls $O/prof_bench $O/prof_serial $O/traffic_FETCH_SIZE | head -20
