#!/bin/bash
# round 3, job 20: fixed-base mode, automatic plan (c by the cost model, staged level A, sliced cooperative level B, one inversion per 8 table entries)
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu --timeout 600 -k "fixed_base or skew or oversized or partition or preloaded" > gpurun_out/r3_20_pytest.log 2>&1
tail -3 gpurun_out/r3_20_pytest.log
timeout 900 python tools/gpu_msm_fixed.py 22:20 23 24 25 26 > gpurun_out/r3_20_fixed.log 2>&1
timeout 300 python tools/gpu_msm_fixed.py bn254 24 26 >> gpurun_out/r3_20_fixed.log 2>&1
grep -v amdgpu gpurun_out/r3_20_fixed.log
timeout 200 python tools/gpu_msm_skew.py 2>&1 | grep -v amdgpu | tail -8
