#!/bin/bash
# round 4, job 30: the one-stage-per-round passes of the 256-bit fields (k_ntt_pass_lat): parity against the oracle
# and the reference's build, then timings with the register passes (SPPARK_NTT_LAT_SMAX=0) and with 8 / 6 stages
mkdir -p gpurun_out; out=gpurun_out/r4_30_ntt_wide_lat.log; : > $out
timeout 900 python -m pytest tests/test_ntt_vs_reference_gpu.py tests/test_ntt_gpu.py -q -x -m gpu -k "bls12_381 or bn254 or bls12_377 or pallas or vesta or lde" 2>&1 | tail -4 | tee -a $out
for s in 0 8 6; do
  echo "== SPPARK_NTT_LAT_SMAX=$s" | tee -a $out
  SPPARK_NTT_LAT_SMAX=$s NTT_FIELDS=bls12_381 timeout 300 python tools/gpu_ntt_vs_reference.py 2>&1 | grep -v amdgpu.ids | tee -a $out
  SPPARK_NTT_LAT_SMAX=$s timeout 300 python tools/gpu_ntt_small_vs_reference.py field=bls12_381 only=ours 2>&1 | grep -v amdgpu.ids | tee -a $out
done
