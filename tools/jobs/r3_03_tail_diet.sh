# round 3, job 3: join with one addition site, quotient digits without masks, no bucket memset; multi-rank dry run,
# in-place LDE_expand, per-device timings (new GPU tests); counters of the radix-64 NTT kernels and of k_accumulate
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r3_03_pytest_gpu.log 2>&1; tail -5 $R/gpurun_out/r3_03_pytest_gpu.log
L=$R/gpurun_out/r3_03_msm_ab.log
echo "=== this build" > $L
timeout 600 python tools/gpu_msm_tail.py sweep 23 26 >> $L 2>&1
timeout 300 python tools/gpu_msm_tail.py ab 16 20 24 25 >> $L 2>&1
echo "=== the build of job 2 (masked quotient digits, two addition sites in k_join_runs, bucket memset)" >> $L
cp sppark_amd/lib/libsppark_bls12_381.so /tmp/new.so; cp tools/exp/ab/libsppark_bls12_381_r3a.so sppark_amd/lib/libsppark_bls12_381.so
timeout 300 python tools/gpu_msm_tail.py ab 16 20 23 24 25 26 >> $L 2>&1
cp /tmp/new.so sppark_amd/lib/libsppark_bls12_381.so
echo "=== this build, again (same box)" >> $L
timeout 300 python tools/gpu_msm_tail.py ab 23 26 >> $L 2>&1
grep -v amdgpu.ids $L
timeout 600 python tools/gpu_msm_tail.py grid 14 16 18 20 > $R/gpurun_out/r3_03_msm_small_grid.log 2>&1; grep "best\|auto" $R/gpurun_out/r3_03_msm_small_grid.log
rm -f $R/gpurun_out/pmc_ntt_gl64.txt $R/gpurun_out/pmc_msm_acc.txt
bash tools/gpu_pmc_job.sh ntt_gl64 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD|FETCH_SIZE|WRITE_SIZE" -- python tools/gpu_ntt_one.py gl64 24 6
bash tools/gpu_pmc_job.sh msm_acc "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" -- python tools/gpu_msm_one.py 26 0
