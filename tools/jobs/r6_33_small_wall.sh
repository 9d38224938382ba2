# Small-size walls again (the box of r6_32 ran 40 % slow at every size).  Outputs: gpurun_out/r6_33_small_wall.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=" > $R/gpurun_out/r6_33_small_wall.log; cat $R/gpurun_out/r6_33_small_wall.log
timeout 300 python tools/gpu_msm_tail.py ab 10 12 14 16 2>&1 | grep -v amdgpu | grep "auto" | tee -a $R/gpurun_out/r6_33_small_wall.log
