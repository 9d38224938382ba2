# round 2, job 15: LDS counters of the two sort kernels
set -x
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_sort_$tag
  (cd $R && timeout 200 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_sort_$tag -o msm -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/pmc_sort_$tag.log 2>&1)
done
cd $R
for d in gpurun_out/pmc_sort_*/; do python tools/rocprof_summary.py $(find $d -name "*.db" | head -1) $(find $d -name "*.db" | head -1) 2>/dev/null | grep "k_sortB\|k_scatterA_staged\|k_histA"; done > gpurun_out/r2_sort_pmc.txt
cat gpurun_out/r2_sort_pmc.txt
rm -rf gpurun_out/pmc_sort_*
