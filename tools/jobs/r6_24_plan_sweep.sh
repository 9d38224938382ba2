# One-knob sweeps around the automatic plan: alt_bn128 G1 at the medium sizes, BLS12-381 G1 at 2^19 / 2^21.  Outputs: gpurun_out/r6_24_plan_sweep.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
(timeout 600 python tools/gpu_msm_tail.py bn254 sweep 16 18 20 22 24; timeout 600 python tools/gpu_msm_tail.py sweep 19 21) 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_24_plan_sweep.log
cut -c1-150 $R/gpurun_out/r6_24_plan_sweep.log
