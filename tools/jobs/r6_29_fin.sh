# The wire image of the window sums from the top's last kernel at every size (no k_finalize): MSM GPU tests, sizes.  Outputs: gpurun_out/r6_29_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_29_pytest_msm.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_29_pytest_msm.log
timeout 400 python tools/gpu_msm_tail.py ab 17 18 19 20 21 22 2>&1 | grep -v amdgpu | grep "auto" > $R/gpurun_out/r6_29_sizes.log; cat $R/gpurun_out/r6_29_sizes.log
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=" > $R/gpurun_out/r6_29_small_wall.log; cat $R/gpurun_out/r6_29_small_wall.log
