#!/bin/bash
# Round 6, job 9: the whole GPU suite on the tree with the shift conversion, the 16-byte point loads of k_convert_points and the
# coset fold through a generic top pass; the headline's kernels (k_convert_points); the G2 plan sweep; the small sizes.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -m gpu -x -q --timeout 900 > $O/r6_09_pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/r6_09_pytest.log | head -5
rm -rf $O/hl; timeout 600 rocprofv3 --kernel-trace --stats -d $O/hl -o hl -- python $R/tools/gpu_msm_one.py 26 0 > $O/r6_09_one_26.log 2>&1
db=$(find $O/hl -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/r6_09_headline_kernels.txt 2>&1; rm -rf $O/hl
head -16 $O/r6_09_headline_kernels.txt | cut -c1-160
timeout 600 python $R/tools/gpu_msm_tail.py ab 12 16 20 23 26 > $O/r6_09_msm_sizes.log 2>&1; grep "auto" $O/r6_09_msm_sizes.log
SPPARK_LIBDIR=lib_tuning timeout 1500 python $R/tools/gpu_g2_sweep.py 22 20 > $O/r6_09_g2_sweep.log 2>&1
grep "G2" $O/r6_09_g2_sweep.log
