# Full-step gradient noise of the cfg-2 step (hip-64 / 32-64 per layer, tools/diag_fullstep.py) for several data / weight
# seeds, product library.  usage: tools/r3/noise_seeds.sh <outfile>
R=$GRAFT_REPO_ROOT; O=${1:-$R/gpurun_out/noise_seeds.txt}
: > $O
for sd in "SEED=1234 WSEED=1" "SEED=77 WSEED=1" "SEED=1234 WSEED=5" "SEED=9 WSEED=11" "SEED=3 WSEED=21" "SEED=50 WSEED=8"; do
  env $sd SUMMARY="$sd" timeout -k 10 600 python3 $R/tools/diag_fullstep.py 2>&1 | grep -E "SUMMARY|g_disp0|disp out" >> $O
done
cat $O
