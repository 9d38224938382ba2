#!/bin/bash
# round 3, job 15: fixed-base one-window MSM with 2^12 level-A partitions (staged scatter): c = 26 and c = 24, timelines
set -x
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in 26:13 24:11 24:12; do
  wb=${cfg%%:*}; lb=${cfg##*:}
  (cd $R && FB_LB=$lb timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_fixed.py --only-fixed 26:$wb > $R/gpurun_out/r3_15_tl_$wb_$lb.log 2>&1)
  grep "fixed-base" $R/gpurun_out/r3_15_tl_$wb_$lb.log
  (cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 34 > gpurun_out/r3_15_fixed_timeline_c${wb}_lb${lb}.txt 2>&1)
  grep -v "big_\|fillBuffer\|reduce_runs\|copyBuffer" $R/gpurun_out/r3_15_fixed_timeline_c${wb}_lb${lb}.txt | tail -22 | cut -c1-130
  rm -rf $R/gpurun_out/prof_tl
done
