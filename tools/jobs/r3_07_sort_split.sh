# round 3, job 7: the automatic sort split (2^14-entry partitions) and slab count -- parity, then all sizes
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 100 > $R/gpurun_out/r3_07_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r3_07_pytest_msm.log
timeout 400 python tools/gpu_msm_tail.py ab 12 14 16 18 19 20 21 22 23 24 25 26 > $R/gpurun_out/r3_07_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_07_msm_sizes.log | grep auto
timeout 200 python tools/gpu_msm_tail.py bn254 ab 16 20 23 26 > $R/gpurun_out/r3_07_msm_sizes_bn254.log 2>&1; grep auto $R/gpurun_out/r3_07_msm_sizes_bn254.log
timeout 200 python tools/gpu_g2_bench.py > $R/gpurun_out/r3_07_g2.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_07_g2.log | tail -6
