#!/bin/bash
# Round 5, job 13 (the review's item 7, measured): level-A sort records of 4 bytes, the index bits above IB recovered in level B
# from the record's position (msm_sort_records.hpp).  The MSM GPU tests first (both record widths run: explicit slab counts,
# the fixed-base mode and the chunked path keep 8 bytes), then the A/B on one box, then rocprofv3 kernel times of 2^26 MSMs.
R=$PWD; mkdir -p $R/gpurun_out; out=$R/gpurun_out/r5_13
timeout 1500 python -m pytest tests/test_msm_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $out.tests.log
timeout 400 python tools/gpu_msm_records_ab.py 26 24 22 20 16 2>&1 | grep "^2\^\|Error\|error" | tee $out.records_ab.log
cd /tmp && export TMPDIR=/tmp
for mode in packed wide; do
  rm -rf $R/gpurun_out/prof_r5r
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5r -o msm -- python tools/gpu_msm_records_one.py $mode > /dev/null 2>&1)
  echo "== $mode records: rocprofv3 kernel stats of 3 MSMs of 2^26 points" | tee -a $out.records_ab.log
  (cd $R && python tools/rocprof_summary.py $(find gpurun_out/prof_r5r -name "*.db" | head -1) 2>&1 | grep -i "kernel \|scatterA\|sortB\|histA\|breakdown\|convert\|accumulate\|scan_" | head -12 | cut -c1-130 | tee -a $out.records_ab.log)
  rm -rf $R/gpurun_out/prof_r5r
done
