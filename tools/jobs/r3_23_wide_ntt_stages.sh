#!/bin/bash
# round 3, job 23: 256-bit NTT with 5 / 6 stages per pass (fewer passes; 2 waves per SIMD) against the 4-stage default, by tile shape
set -x
mkdir -p gpurun_out
: > gpurun_out/r3_23_wide_stages.log
for spec in 4:4:10 5:4:10 5:4:11 5:3:10 6:4:10 6:4:11 6:4:12 6:3:10 6:3:11 6:2:10 6:2:11 5:2:10; do
  timeout 120 python tools/gpu_ntt_wide_knobs.py 24 $spec 2>&1 | grep "bls12_381" >> gpurun_out/r3_23_wide_stages.log
done
cat gpurun_out/r3_23_wide_stages.log
SPPARK_NTT_SMAX=6 timeout 600 python -m pytest tests/test_ntt_gpu.py -x -q -m gpu --timeout 600 -k "wide or bls12 or bn254 or lde or full_size" > gpurun_out/r3_23_pytest_s6.log 2>&1; tail -3 gpurun_out/r3_23_pytest_s6.log
