cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=${LAYER:-conv3.2}
for d in $DBGS; do
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  DVF_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/pmcx_${d}_$i -- python3 $R/tools/prof_one.py $L fwd > /dev/null 2>&1
done; done
find $R/gpurun_out -path "*pmcx_*" -name "*counter_collection.csv" | head
