#!/bin/bash
# Round 5, job 3: alt_bn128 G1 on nine 29-bit limbs (ff/montx_dev.hpp TIGHT) -- parity of every alt_bn128 GPU test, then the
# A/B against the ten-limb build (SPPARK_LIBDIR=lib_lb28, -DSPPARK_BN254_LB=28) on the same box: 2^26 / 2^22 / 2^16.
mkdir -p gpurun_out; out=gpurun_out/r5_03
timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_field_vs_reference_gpu.py -q -x -m gpu -k "bn254 or 1-bn254 or bucket or point_ops or field_ops" --timeout 300 2>&1 | tail -6 | tee $out.pytest.log
for lib in lib_lb28 lib; do
  echo "== $lib" | tee -a $out.ab.log
  for lg in 26 22 16; do
    SPPARK_LIBDIR=$lib timeout 200 python tools/gpu_msm_bn254.py $lg 2>&1 | grep -v amdgpu | tail -2 | tee -a $out.ab.log
  done
done
