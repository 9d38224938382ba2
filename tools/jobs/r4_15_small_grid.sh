# round 4, job 15: window / run-length grid at 2^12 .. 2^18 with the cooperative tail (the optimum of round 3 was found with the old tail costs)
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python tools/gpu_msm_tail.py grid 12 14 15 16 17 18 > $R/gpurun_out/r4_15_small_grid.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_15_small_grid.log | grep "best\|auto  " 
