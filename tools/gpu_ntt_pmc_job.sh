R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/pmc_avail.txt 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_ntt_$tag
  (cd $R && timeout 200 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_ntt_$tag -o ntt -- python tools/gpu_ntt_one.py ${NTT_FIELD:-gl64} 24 6 > $R/gpurun_out/pmc_ntt_$tag.log 2>&1)
  tail -2 $R/gpurun_out/pmc_ntt_$tag.log | cut -c1-200
done
ls $R/gpurun_out | grep pmc_ntt
