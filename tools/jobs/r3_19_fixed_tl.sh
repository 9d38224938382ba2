#!/bin/bash
# round 3, job 19: timeline of the fixed-base MSM at 2^26 (c = 24) with the sliced cooperative level B
set -x
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_fixed.py --only-fixed 26:24 > $R/gpurun_out/r3_19_tl.log 2>&1)
(cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 30 > gpurun_out/r3_19_fixed_timeline_2p26.txt 2>&1)
grep -v "fillBuffer\|reduce_runs\|copyBuffer" $R/gpurun_out/r3_19_fixed_timeline_2p26.txt | tail -26 | cut -c1-130
rm -rf $R/gpurun_out/prof_tl
