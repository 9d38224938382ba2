#!/bin/bash
# Round 6, job 15: the window sums' wire image written by k_bucket_top_sum_coop (no k_finalize launch) on the small windows' path:
# MSM GPU tests, small sizes with the phase timers on and off.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 2400 python -m pytest $R/tests/test_msm_gpu.py -x -q --timeout 900 > $O/r6_15_pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/r6_15_pytest.log | head -5
timeout 600 python $R/tools/gpu_msm_tail.py ab 10 12 14 15 16 > $O/r6_15_msm_sizes.log 2>&1; grep "auto" $O/r6_15_msm_sizes.log
timeout 300 python $R/tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=False" | tee $O/r6_15_wall_timers_off.log
