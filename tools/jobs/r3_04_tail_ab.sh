# round 3, job 4: the build with 238 registers in k_accumulate again (old zero test), products in pairs at the top of
# the bucket sums, L = 128 / 256, K1 = 16; tail sweep, small-size grid, host-buffer path, SQ counters of k_accumulate
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r3_04_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r3_04_pytest_msm.log
L=$R/gpurun_out/r3_04_msm_tail.log
timeout 600 python tools/gpu_msm_tail.py sweep 23 26 > $L 2>&1
timeout 300 python tools/gpu_msm_tail.py ab 16 20 22 24 25 >> $L 2>&1
grep -v amdgpu.ids $L
timeout 600 python tools/gpu_msm_tail.py grid 14 16 18 > $R/gpurun_out/r3_04_msm_small_grid.log 2>&1; grep "best\|auto" $R/gpurun_out/r3_04_msm_small_grid.log
timeout 600 python tools/gpu_msm_host.py 24 26 > $R/gpurun_out/r3_04_msm_host.log 2>&1; grep -v amdgpu.ids $R/gpurun_out/r3_04_msm_host.log
rm -f $R/gpurun_out/pmc_msm_acc2.txt
bash tools/gpu_pmc_job.sh msm_acc2 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" -- python tools/gpu_msm_one.py 26 0 | grep -i "accumulate\|join\|level\|top\|kernel"
