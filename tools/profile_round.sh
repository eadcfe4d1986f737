# Profiles judged for the round (run on the GPU box through gpurun).  Outputs under gpurun_out/prof/; tools/traffic.py, tools/db2stats.py
# and a few cp's turn them into profiles/<prefix>_*.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o serial -- python3 $R/bench.py --steps 5 --warmup 2 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > $O/bench_serial.json 2> $O/bench_serial.err
echo "serial done"
for c in 3 4 5; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg$c -o cfg$c -- python3 $R/bench.py --config $c --steps 5 --warmup 2 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > $O/bench_cfg${c}_serial.json 2> $O/bench_cfg${c}_serial.err
done
echo "cfg stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --serialize --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  echo "pmc $c done"
done
for c in 3 4 5; do python3 $R/bench.py --config $c --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
DVF_LAYER_TABLE=400 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/layer_table.txt
# per-layer kernel durations from the profiler (not from event brackets): call log of the recorded pass x kernel trace
DVF_CALL_LOG=$O/calls.jsonl rocprofv3 --kernel-trace --output-format csv -d $O/prof_calls -o calls -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --serialize --no-cpu-baseline > /dev/null 2> $O/bench_calls.err
python3 $R/tools/r3/launch_table.py $O/calls.jsonl $(ls $O/prof_calls/*kernel_trace.csv | head -1) > $O/launch_table.txt
tail -12 $O/launch_table.txt
find $O -name "*.csv" | head -30
# keep the merged payload small: drop the per-launch traces, keep stats + counter csv
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
