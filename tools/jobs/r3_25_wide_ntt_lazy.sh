#!/bin/bash
# round 3, job 25: 256-bit NTT passes over the lazily reduced 28-bit-limb class: parity (NTT + LDE + poly suites), timings
set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ntt_gpu.py tests/test_poly_gpu.py -x -q -m gpu --timeout 600 > gpurun_out/r3_25_pytest.log 2>&1
tail -3 gpurun_out/r3_25_pytest.log
NTT_FIELDS=bls12_381,bn254 NTT_LGS=12,16,20,22,24,26 timeout 300 python tools/gpu_ntt_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r3_25_wide_ntt.log
cat gpurun_out/r3_25_wide_ntt.log | cut -c1-230
