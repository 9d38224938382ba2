# Sort split (low bits) x point slabs around the automatic plan at 2^18 .. 2^26 (watch before-acc).  Outputs: gpurun_out/r6_28_sort_sweep.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python tools/gpu_msm_tail.py sort 18 20 22 24 26 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_28_sort_sweep.log
cut -c1-150 $R/gpurun_out/r6_28_sort_sweep.log
