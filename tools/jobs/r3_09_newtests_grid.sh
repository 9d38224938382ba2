# round 3, job 9: the new GPU tests (tail variants, NTT plan knobs in fresh processes) and the small-size (window, run length)
# grid with the final kernels (sort split, join skip, low-latency sums)
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_ntt_gpu.py -m gpu -x -q --timeout 300 -k "tail_variants or plan_knobs or bucket_sum_top or golden" > $R/gpurun_out/r3_09_pytest_new.log 2>&1; tail -3 $R/gpurun_out/r3_09_pytest_new.log
timeout 700 python tools/gpu_msm_tail.py grid 12 14 16 18 19 > $R/gpurun_out/r3_09_msm_small_grid.log 2>&1; grep "best\|auto" $R/gpurun_out/r3_09_msm_small_grid.log
