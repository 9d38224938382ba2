#!/bin/bash
# Round 5, job 12: k_ntt_small as kept after job 11 (twiddle rows of ntt_tables::inner, one instance per size 2^8 ... 2^11, one
# pair per lane throughout): the NTT / polynomial GPU tests, then all four orders against the reference's build, 2^8 ... 2^20.
mkdir -p gpurun_out; out=gpurun_out/r5_12; : > $out.ntt_timing.log
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out.tests.log
for o in 1 0 2 3; do
  echo "== order $o" | tee -a $out.ntt_timing.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep "^gl64\|^bb31\|^bls12_381\|^bn254\|rows" | grep "2^8 \|2^9 \|2^10 \|2^11 \|2^12 \|2^16 \|2^20 \|rows" | tee -a $out.ntt_timing.log
done
