# (window bits x run length) grid of the small sizes for the other curves' fields (alt_bn128: nine 29-bit limbs; BLS12-377: fourteen 28-bit).  Outputs: gpurun_out/r6_30_small_grid_curves.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
(for c in bn254 bls12_377; do timeout 600 python tools/gpu_msm_tail.py $c grid 12 14 16 2>&1 | grep -v amdgpu | sed "s/^/$c /"; done) > $R/gpurun_out/r6_30_small_grid_curves.log
grep "best\|auto " $R/gpurun_out/r6_30_small_grid_curves.log | cut -c1-160
