#!/bin/bash
# Round 6, job 6: the piece tree with the pair-major mapping and the new first-level chunks: sizes A/B, the kernel timeline of
# 2^12 / 2^16, the bucket-sum knobs at 2^15 / 2^17.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/gpu_msm_tail.py ab 10 12 14 15 16 17 18 19 20 > $O/r6_06_msm_sizes.log 2>&1; grep "auto\|no piece" $O/r6_06_msm_sizes.log
for lg in 12 16; do
  rm -rf $O/tl_$lg
  timeout 300 rocprofv3 --kernel-trace -d $O/tl_$lg -o tl -- python $R/tools/gpu_msm_one.py $lg 0 > $O/r6_06_one_$lg.log 2>&1
  db=$(find $O/tl_$lg -name "*.db" | head -1)
  python $R/tools/rocprof_timeline.py $db 30 > $O/r6_06_timeline_2p$lg.txt 2>&1
  rm -rf $O/tl_$lg
done
tail -26 $O/r6_06_timeline_2p16.txt | cut -c1-150
timeout 600 python $R/tools/gpu_msm_sums_sweep.py 15 17 > $O/r6_06_sums_sweep.log 2>&1
for lg in 15 17; do grep "^2^$lg auto" $O/r6_06_sums_sweep.log; grep "^2^$lg K1" $O/r6_06_sums_sweep.log | awk '{print $NF, $0}' | sort -n | head -3 | cut -d' ' -f2-; done
