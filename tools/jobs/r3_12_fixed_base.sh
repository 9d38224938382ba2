#!/bin/bash
# round 3, job 12: fixed-base tables -- parity tests, then timings against the plain preloaded path
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu --timeout 600 -k "fixed_base or preloaded" > gpurun_out/r3_12_pytest.log 2>&1
tail -5 gpurun_out/r3_12_pytest.log
timeout 600 python tools/gpu_msm_fixed.py 16 18 20:0,16,18 22 > gpurun_out/r3_12_fixed.log 2>&1
tail -40 gpurun_out/r3_12_fixed.log
