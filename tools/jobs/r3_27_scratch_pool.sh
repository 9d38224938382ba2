#!/bin/bash
# round 3, job 27: device scratch of compute_ntt (host buffers) / sppark_lde / the polynomial scans kept between calls: parity + timings
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_poly_gpu.py -x -q -m gpu --timeout 300 > gpurun_out/r3_27_pytest.log 2>&1; tail -2 gpurun_out/r3_27_pytest.log
for spec in "gl64 22 2" "gl64 20 3" "bb31 22 2" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> gpurun_out/r3_27_lde.log; done
timeout 120 python tools/gpu_poly_one.py 2>&1 | grep -v amdgpu >> gpurun_out/r3_27_lde.log
cat gpurun_out/r3_27_lde.log
