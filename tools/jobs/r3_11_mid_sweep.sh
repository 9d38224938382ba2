# round 3, job 11: one-knob sweeps around the automatic plan at 2^20 .. 2^22 with the final kernels
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python tools/gpu_msm_tail.py sweep 19 20 21 22 > $R/gpurun_out/r3_11_msm_mid_sweep.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_11_msm_mid_sweep.log
