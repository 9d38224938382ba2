#!/bin/bash
# round 3, job 17: cooperative level-B with slices of ~8-16 K entries (runtime split, 4 loads in flight); parity of the skewed cases, then fixed-base timings
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -x -q -m gpu --timeout 600 -k "fixed_base or skew or oversized or partition or big" > gpurun_out/r3_17_pytest.log 2>&1
tail -3 gpurun_out/r3_17_pytest.log
timeout 600 python tools/gpu_msm_fixed.py --only-fixed 26 26:26 > gpurun_out/r3_17_fixed.log 2>&1
FB_BIG=32768 timeout 600 python tools/gpu_msm_fixed.py 22 24 >> gpurun_out/r3_17_fixed.log 2>&1
timeout 600 python tools/gpu_msm_fixed.py --only-fixed 22 24 >> gpurun_out/r3_17_fixed.log 2>&1
grep -v amdgpu gpurun_out/r3_17_fixed.log
