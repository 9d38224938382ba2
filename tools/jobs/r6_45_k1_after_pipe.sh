# First-level chunk of the bucket sums at the medium sizes again, now that the level runs its two sums on two waves.  Outputs: gpurun_out/r6_45_k1.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python tools/gpu_msm_tail.py sweep 17 18 19 20 21 2>&1 | grep -v amdgpu | grep "auto\|K1=" | tee $R/gpurun_out/r6_45_k1.log
