# round 4, job 2: the 16-byte bit reversal (NTT tests + timings), the point conversion beside the sort (A/B in separate processes)
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_ntt_gpu.py -m gpu -x -q > $R/gpurun_out/r4_02_pytest_ntt.log 2>&1; tail -3 $R/gpurun_out/r4_02_pytest_ntt.log
timeout 300 env NTT_FIELDS=gl64,bb31 NTT_LGS=13,16,20,24,26 python tools/gpu_ntt_bench.py > $R/gpurun_out/r4_02_ntt_bench.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_02_ntt_bench.log
for aux in 0 1 0 1; do
  echo "SPPARK_MSM_CONVERT_AUX=$aux" >> $R/gpurun_out/r4_02_convert_ab.log
  timeout 300 env SPPARK_MSM_CONVERT_AUX=$aux python tools/gpu_msm_tail.py sort 26 24 22 20 2>&1 | grep "auto" >> $R/gpurun_out/r4_02_convert_ab.log
done
cat $R/gpurun_out/r4_02_convert_ab.log
cd /tmp && export TMPDIR=/tmp; cd $R
bash tools/gpu_pmc_job.sh bitrev_vec "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" -- python tools/gpu_ntt_one.py bb31 24 6 NN | grep -i "bitrev\|kernel "
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "full_size or pipeline_medium or window_groups or golden" > $R/gpurun_out/r4_02_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r4_02_pytest_msm.log
