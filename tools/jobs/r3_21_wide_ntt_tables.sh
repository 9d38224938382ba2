#!/bin/bash
# round 3, job 21: 256-bit NTT with inter-pass twiddle TABLES (one product per element and pass saved): parity, then A/B by table limit
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ntt_gpu.py -x -q -m gpu --timeout 600 -k "wide or bls12 or bn254 or fr or lde or full_size" > gpurun_out/r3_21_pytest.log 2>&1
tail -3 gpurun_out/r3_21_pytest.log
for lim in 0 16 20 24; do
  SPPARK_NTT_WIDE_TABLE=$lim NTT_FIELDS=bls12_381,bn254 NTT_LGS=16,20,22,24,26 timeout 300 python tools/gpu_ntt_bench.py 2>&1 | grep -v amdgpu >> gpurun_out/r3_21_wide_ntt.log
done
cat gpurun_out/r3_21_wide_ntt.log | cut -c1-230
