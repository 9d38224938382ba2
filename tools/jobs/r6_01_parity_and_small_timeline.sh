#!/bin/bash
# Round 6, job 1: the new parity legs (odd prime point periods, the all-distinct 2^26 vector, the progression generator)
# and, for the small-MSM work, the kernel timeline of one MSM at 2^12 / 2^16 / 2^20 plus the host-side overhead.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests/test_msm_gpu.py -x -q --timeout 600 \
  -k "progression or all_distinct or full_size or above_2p28 or skewed_scalars_full or pipeline_medium" > $O/r6_01_pytest.log 2>&1
tail -5 $O/r6_01_pytest.log
for lg in 12 16 20; do
  rm -rf $O/tl_$lg
  timeout 300 rocprofv3 --kernel-trace -d $O/tl_$lg -o tl -- python $R/tools/gpu_msm_one.py $lg 0 > $O/r6_01_one_$lg.log 2>&1
  db=$(find $O/tl_$lg -name "*.db" | head -1)
  python $R/tools/rocprof_timeline.py $db 70 > $O/r6_01_timeline_2p$lg.txt 2>&1
  tail -3 $O/r6_01_timeline_2p$lg.txt
  rm -rf $O/tl_$lg
done
timeout 300 python $R/tools/gpu_msm_tail.py ab 10 12 14 16 18 20 > $O/r6_01_msm_sizes.log 2>&1; cat $O/r6_01_msm_sizes.log | grep auto
