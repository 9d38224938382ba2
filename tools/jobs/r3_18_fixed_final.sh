#!/bin/bash
# round 3, job 18: fixed-base mode with the staged level A + sliced cooperative level B: parity, then 2^23..2^26 against the plain preloaded path
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu --timeout 600 -k "fixed_base or skew or oversized or partition or preloaded" > gpurun_out/r3_18_pytest.log 2>&1
tail -3 gpurun_out/r3_18_pytest.log
timeout 900 python tools/gpu_msm_fixed.py 23:22,23 24:22,24 25:24 26:24,26 > gpurun_out/r3_18_fixed.log 2>&1
grep -v amdgpu gpurun_out/r3_18_fixed.log
