#!/bin/bash
# Round 5, job 4: the register-exchange form of k_ntt_small (parity: every NTT / LDE / polynomial GPU test), the Pasta curves
# on nine 29-bit limbs (parity), then the small-transform timing against the reference's build, NR and NN orders.
mkdir -p gpurun_out; out=gpurun_out/r5_04
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -q -x -m gpu --timeout 300 2>&1 | tail -6 | tee $out.pytest_ntt.log
timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_field_vs_reference_gpu.py -q -x -m gpu -k "pallas or vesta" --timeout 300 2>&1 | tail -6 | tee $out.pytest_pasta.log
timeout 300 python tools/gpu_ntt_small_vs_reference.py 2>&1 | grep -v amdgpu | tee $out.ntt_small.log
for o in 0 2 3; do
  echo "== order $o" | tee -a $out.ntt_small.log
  timeout 200 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|rows" | tee -a $out.ntt_small.log
done
