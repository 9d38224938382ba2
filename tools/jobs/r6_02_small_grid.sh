#!/bin/bash
# Round 6, job 2: (window bits x run length) grid of the small sizes with the CURRENT tail kernels (the automatic plan below 2^19
# dates from rounds 1-2, before the join kernel and the cooperative forms existed).
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python $R/tools/gpu_msm_tail.py grid 10 12 14 16 17 18 19 > $O/r6_02_small_grid.log 2>&1
grep -n "best\|auto" $O/r6_02_small_grid.log
