# round 3, job 6: k_reduce_tail with 256 lanes (the 1024-lane build hung the G2 tests: every pytest run now has a
# per-test timeout), then the split of the bucket index between the two sort levels and the slab count at mid sizes
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 100 > $R/gpurun_out/r3_06_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r3_06_pytest_msm.log
timeout 300 python tools/gpu_msm_tail.py ab 14 16 18 20 > $R/gpurun_out/r3_06_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_06_msm_sizes.log
timeout 500 python tools/gpu_msm_tail.py sort 16 18 20 22 23 24 > $R/gpurun_out/r3_06_msm_sort_split.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_06_msm_sort_split.log
