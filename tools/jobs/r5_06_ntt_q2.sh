#!/bin/bash
# Round 5, job 6: k_ntt_small with two butterfly pairs per lane from 2^10 on (default build): parity, timing against the
# reference's build; then the tuning build with the switch at 8 / 9 / 10 / never, and the 256-bit fields at 2^10 with it.
mkdir -p gpurun_out; out=gpurun_out/r5_06
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -q -x -m gpu --timeout 300 2>&1 | tail -4 | tee $out.pytest_ntt.log
for o in 1 2; do
  echo "== default build, order $o" | tee -a $out.ntt_small.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|2^11 \|rows" | tee -a $out.ntt_small.log
done
for q2 in 8 9 10 99; do
  echo "== tuning build, SPPARK_NTT_SMALL_Q2=$q2 (ours only), order 1" | tee -a $out.ntt_small.log
  SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_Q2=$q2 timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours 2>&1 | grep -v amdgpu | grep "^gl64\|^bb31" | grep "2^8 \|2^9 \|2^10 \|2^11 " | tee -a $out.ntt_small.log
done
echo "== tuning build, 256-bit at 2^10 by k_ntt_small: one pair per lane / two pairs per lane" | tee -a $out.ntt_small.log
SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_MAX=10 SPPARK_NTT_SMALL_Q2=99 timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours field=bls12_381 2>&1 | grep "2^9 \|2^10 " | tee -a $out.ntt_small.log
SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_MAX=10 SPPARK_NTT_SMALL_Q2=9 timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours field=bls12_381 2>&1 | grep "2^9 \|2^10 " | tee -a $out.ntt_small.log
