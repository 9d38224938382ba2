#!/bin/bash
# Round 5, job 8: k_ntt_small with the twiddles of the upper stages by squaring instead of table gathers (single-word fields):
# parity, then all four orders against the reference's build.
mkdir -p gpurun_out; out=gpurun_out/r5_08; : > $out.ntt_small.log
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -q -x -m gpu --timeout 300 2>&1 | tail -4 | tee $out.pytest_ntt.log
for o in 1 0 2 3; do
  echo "== order $o" | tee -a $out.ntt_small.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|2^11 \|2^12 \|2^16 \|2^20 \|rows" | tee -a $out.ntt_small.log
done
