#!/bin/bash
# Round 6, job 3: bucket-sum knobs (K1, K, top hand-over) at 2^12 .. 2^20
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python $R/tools/gpu_msm_sums_sweep.py 12 14 16 18 20 > $O/r6_03_sums_sweep.log 2>&1
tail -3 $O/r6_03_sums_sweep.log
