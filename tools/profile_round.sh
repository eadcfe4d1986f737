# Profiles judged for the round (run on the GPU box through gpurun).  Outputs under gpurun_out/; tools/traffic.py and a
# few cp's turn them into profiles/r02_*.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats -d $O/prof_serial -o serial -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > $O/bench_serial.json 2> $O/bench_serial.err
rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 -o cfg3 -- python3 $R/bench.py --config 3 --steps 5 --warmup 2 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > $O/bench_cfg3_serial.json 2> $O/bench_cfg3_serial.err
rocprofv3 --kernel-trace --stats -d $O/prof_cfg5 -o cfg5 -- python3 $R/bench.py --config 5 --steps 5 --warmup 2 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > $O/bench_cfg5_serial.json 2> $O/bench_cfg5_serial.err
rocprofv3 --kernel-trace --stats -d $O/prof_photo -o photo -- python3 $R/tools/photo_bench.py 20 > $O/photo_bench.txt 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic3_$c -- python3 $R/bench.py --config 3 --steps 3 --warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
done
for c in 3 4 5; do python3 $R/bench.py --config $c --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
ls $O/prof_bench $O/prof_serial $O/traffic_FETCH_SIZE | head -20
