#!/bin/bash
# round 3, job 24: G2 pipeline over the 28-bit-limb base field (fp2x_dev): parity, then timings (round-2/3 figure with fp2_dev: 2^22 in 66 ms)
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu --timeout 600 -k "g2 or fp2" > gpurun_out/r3_24_pytest.log 2>&1
tail -3 gpurun_out/r3_24_pytest.log
timeout 600 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r3_24_g2.log
cat gpurun_out/r3_24_g2.log
