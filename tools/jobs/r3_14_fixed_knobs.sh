#!/bin/bash
# round 3, job 14: the one-window MSM of the fixed-base mode -- sort split / slabs / run length / first bucket level, and its timeline at 2^26
set -x
R=$PWD; mkdir -p $R/gpurun_out
FB_LB=9,11,12,13 FB_SLABS=64,256,2048 FB_L=128 FB_K1=8,32 timeout 900 python tools/gpu_msm_fixed.py --only-fixed 26 > gpurun_out/r3_14_fixed_knobs.log 2>&1
grep -v amdgpu gpurun_out/r3_14_fixed_knobs.log | tail -20
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_fixed.py --only-fixed 26 > $R/gpurun_out/r3_14_tl.log 2>&1)
(cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 40 > gpurun_out/r3_14_fixed_timeline_2p26.txt 2>&1)
tail -44 $R/gpurun_out/r3_14_fixed_timeline_2p26.txt | cut -c1-130
rm -rf $R/gpurun_out/prof_tl
