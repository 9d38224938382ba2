# round 3, job 8: low-latency variants of the bucket-sum levels for small grids (k_bucket_level1_lat / _levelN_lat)
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 100 > $R/gpurun_out/r3_08_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r3_08_pytest_msm.log
timeout 400 python tools/gpu_msm_tail.py ab 12 14 16 18 20 21 22 23 24 > $R/gpurun_out/r3_08_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_08_msm_sizes.log
