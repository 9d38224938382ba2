# round 3, job 10: windows of 14 / 15 / 16 bits at 2^17 / 2^18 / 2^19 -- parity, the grid at 2^17, all sizes
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 100 > $R/gpurun_out/r3_10_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r3_10_pytest_msm.log
timeout 300 python tools/gpu_msm_tail.py grid 17 > $R/gpurun_out/r3_10_msm_grid17.log 2>&1; grep "best\|auto" $R/gpurun_out/r3_10_msm_grid17.log
timeout 300 python tools/gpu_msm_tail.py ab 15 16 17 18 19 20 21 > $R/gpurun_out/r3_10_msm_sizes.log 2>&1; grep auto $R/gpurun_out/r3_10_msm_sizes.log
timeout 200 python tools/gpu_msm_tail.py bn254 ab 16 17 18 19 20 > $R/gpurun_out/r3_10_msm_sizes_bn254.log 2>&1; grep auto $R/gpurun_out/r3_10_msm_sizes_bn254.log
