# Which kernel family injects the extra gradient noise of the decoder (full-size cfg-2 step, hip-64 / 32-64 per layer)?
# One oracle run (fp32 + fp64 on the CPU), then the HIP step under tuning knobs.  usage: tools/r3/noise_knobs.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-noise}; mkdir -p $O
export DVF_LIB=$R/depth-vo-feat_amd/dvf/libdvf_hip_tuning.so ORACLE_CACHE=/tmp/oracle_cfg2.pt
run() { tag=$1; shift; env "$@" SUMMARY=$tag timeout -k 10 600 python3 $R/tools/diag_fullstep.py > $O/noise_$tag.txt 2>&1; grep SUMMARY $O/noise_$tag.txt; }
run base A=1
run nohead DVF_NO_HEAD=1
run nofuse FUSE=0
run det DET=1
run ser SER=1
run nowgpipe DVF_WG_PIPE=0
run noblk DVF_PIPE_BLK=0
run nodconvt DVF_NO_DCONVT=1 DVF_NO_WIDE_HEAD=1
run blk1 DVF_PIPE_BLK=1
