R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_msm_$tag
  (cd $R && timeout 200 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_msm_$tag -o msm -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/pmc_msm_$tag.log 2>&1)
  tail -1 $R/gpurun_out/pmc_msm_$tag.log | cut -c1-150
done
