# One-knob sweeps around the automatic plan, BLS12-381 G1 at 2^22 .. 2^25.  Outputs: gpurun_out/r6_27_plan_sweep_large.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python tools/gpu_msm_tail.py sweep 22 23 24 25 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_27_plan_sweep_large.log
grep -v "join off\|no \|per-lane\|top per" $R/gpurun_out/r6_27_plan_sweep_large.log | cut -c1-150
