#!/bin/bash
# Round 6, job 11: the whole GPU suite with the small windows' bucket sums (k_bucket_small_bits_coop); sizes A/B
# (tune_tail 3 = the chunked first level + subset-sum top as before); timeline of 2^12 / 2^16.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -m gpu -x -q --timeout 900 > $O/r6_11_pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/r6_11_pytest.log | head -5
timeout 600 python $R/tools/gpu_msm_tail.py ab 10 12 14 15 16 17 18 > $O/r6_11_msm_sizes.log 2>&1; grep "auto\|no low-lat\|no piece" $O/r6_11_msm_sizes.log
for lg in 12 16; do
  rm -rf $O/tl_$lg
  timeout 300 rocprofv3 --kernel-trace -d $O/tl_$lg -o tl -- python $R/tools/gpu_msm_tail.py ab $lg > $O/r6_11_one_$lg.log 2>&1
  db=$(find $O/tl_$lg -name "*.db" | head -1)
  python $R/tools/rocprof_timeline.py $db 400 > $O/r6_11_timeline_2p$lg.txt 2>&1
  rm -rf $O/tl_$lg
done
grep -n "k_breakdown" $O/r6_11_timeline_2p16.txt | head -3
