#!/bin/bash
# Round 5, job 5: k_ntt_small with lane-permute regrouping (v_permlane32/16_swap, DPP) and 2^11 capacity: parity, then
# timing against the reference's build in all four orders; A/B of the 2^11 limit (tuning build, SPPARK_NTT_SMALL_MAX=10)
# and of the 256-bit fields at 2^10 (SPPARK_NTT_SMALL_MAX=10 against the default 9).
mkdir -p gpurun_out; out=gpurun_out/r5_05
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -q -x -m gpu --timeout 300 2>&1 | tail -4 | tee $out.pytest_ntt.log
for o in 1 0 2 3; do
  echo "== order $o" | tee -a $out.ntt_small.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|2^11 \|2^12 \|2^16 \|2^20 \|rows" | tee -a $out.ntt_small.log
done
echo "== tuning build, SPPARK_NTT_SMALL_MAX=10 (2^11 by the general path; 256-bit 2^10 by k_ntt_small)" | tee -a $out.ntt_small.log
for o in 1 2; do
SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_MAX=10 timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours order=$o 2>&1 | grep -v amdgpu | grep "2^10 \|2^11 " | tee -a $out.ntt_small.log
done
