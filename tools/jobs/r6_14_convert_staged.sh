#!/bin/bash
# Round 6, job 14: k_convert_points with coalesced accesses through LDS (tune_tail 6 = one lane per point as before): MSM GPU tests,
# then the A/B at 2^22 ... 2^26 and the kernel's own time in a trace of the headline.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests/test_msm_gpu.py -x -q --timeout 900 > $O/r6_14_pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/r6_14_pytest.log | head -5
timeout 600 python $R/tools/gpu_msm_tail.py ab 20 22 24 26 > $O/r6_14_msm_sizes.log 2>&1; grep "auto\|per-lane" $O/r6_14_msm_sizes.log
rm -rf $O/hl; timeout 600 rocprofv3 --kernel-trace --stats -d $O/hl -o hl -- python $R/tools/gpu_msm_tail.py ab 26 > $O/r6_14_one_26.log 2>&1
db=$(find $O/hl -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/r6_14_headline_kernels.txt 2>&1; rm -rf $O/hl
grep "convert" $O/r6_14_headline_kernels.txt | cut -c1-160
