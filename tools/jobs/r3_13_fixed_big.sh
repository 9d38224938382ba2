#!/bin/bash
# round 3, job 13: fixed-base tables at the large sizes + a kernel trace of the fixed-base MSM at 2^22
set -x
mkdir -p gpurun_out
timeout 900 python tools/gpu_msm_fixed.py 24 26 > gpurun_out/r3_13_fixed_big.log 2>&1
tail -20 gpurun_out/r3_13_fixed_big.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof13 -o fb22 -- python $GRAFT_REPO_ROOT/tools/gpu_msm_fixed.py --only-fixed 22 > $GRAFT_REPO_ROOT/gpurun_out/r3_13_trace_run.log 2>&1
find /tmp/prof13 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r3_13_fb22_kernel_stats.csv \;
head -30 $GRAFT_REPO_ROOT/gpurun_out/r3_13_fb22_kernel_stats.csv | cut -c1-160
