#!/bin/bash
# round 3, job 26: sppark_lde with its scratch kept between calls (no hipMalloc / hipFree per call): parity + timings
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ntt_gpu.py -x -q -m gpu --timeout 300 -k "lde" > gpurun_out/r3_26_pytest.log 2>&1; tail -2 gpurun_out/r3_26_pytest.log
for spec in "gl64 22 2" "gl64 20 3" "gl64 24 1" "bb31 22 2" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> gpurun_out/r3_26_lde.log; done
cat gpurun_out/r3_26_lde.log
