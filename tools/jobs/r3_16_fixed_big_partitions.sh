#!/bin/bash
# round 3, job 16: fixed-base one-window MSM, level-B in ONE work-group per partition (two-pass form) instead of the cooperative kernels
set -x
mkdir -p gpurun_out
for cfg in 26:13 25:12,13 24:11,12; do
  wb=${cfg%%:*}; lb=${cfg##*:}
  FB_BIG=4194304 FB_LB=$lb timeout 300 python tools/gpu_msm_fixed.py --only-fixed 26:$wb 2>&1 | grep "fixed-base" >> gpurun_out/r3_16_fixed_big_partitions.log
done
cat gpurun_out/r3_16_fixed_big_partitions.log
