# round 2, job 25: SQ counters of k_accumulate (instructions per wave, issue rate) and of the gl64 NTT pass
set -x
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_acc_$tag
  (cd $R && timeout 200 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_acc_$tag -o msm -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/pmc_acc_$tag.log 2>&1)
done
cd $R
for d in gpurun_out/pmc_acc_*/; do python tools/rocprof_summary.py $(find $d -name "*.db" | head -1) $(find $d -name "*.db" | head -1) 2>/dev/null | grep "k_accumulate\|k_bucket_level1\|k_reduce_runs" | grep "SQ_"; done > gpurun_out/r02_msm_accumulate_sq_pmc.txt
cat gpurun_out/r02_msm_accumulate_sq_pmc.txt
rm -rf gpurun_out/pmc_acc_*
