#!/bin/bash
# Round 5, job 7: two butterfly pairs per lane in k_ntt_small -- from which size?  Tuning build (the one of job 6 had failed to
# compile and measured a stale library), SPPARK_NTT_SMALL_Q2 = 8 / 9 / 10 / 11 / never on ONE box, NR order; then the default
# build (Q2 = 11) against the reference's build in NR and RN order.
mkdir -p gpurun_out; out=gpurun_out/r5_07; : > $out.ntt_small.log
for q2 in 8 9 10 11 99; do
  echo "== tuning build, SPPARK_NTT_SMALL_Q2=$q2 (ours only), order 1" | tee -a $out.ntt_small.log
  SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_Q2=$q2 timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours 2>&1 | grep -v amdgpu | grep "^gl64\|^bb31" | grep "2^8 \|2^9 \|2^10 \|2^11 " | tee -a $out.ntt_small.log
done
for o in 1 2; do
  echo "== default build, order $o" | tee -a $out.ntt_small.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|2^11 \|rows" | tee -a $out.ntt_small.log
done
