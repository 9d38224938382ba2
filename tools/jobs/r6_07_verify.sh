#!/bin/bash
# Round 6, job 7: the whole GPU suite on the tree with the shift conversion (montx_dev::from_std) and the coset fold through a
# generic top pass; then the coset A/B at sizes with a generic top pass, k_convert_points in a kernel trace of the headline,
# the small sizes.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -m gpu -x -q --timeout 900 > $O/r6_07_pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/r6_07_pytest.log | head -5
{
for fold in 1 0 1 0; do
  SPPARK_LIBDIR=lib_tuning SPPARK_NTT_COSET_FOLD=$fold NTT_LGS=16,20,22,26 timeout 300 python $R/tools/gpu_ntt_orders.py
done
} > $O/r6_07_ntt_orders_generic_top_ab.log 2>&1
grep "coset" $O/r6_07_ntt_orders_generic_top_ab.log | cut -c1-150
rm -rf $O/hl; timeout 600 rocprofv3 --kernel-trace --stats -d $O/hl -o hl -- python $R/tools/gpu_msm_one.py 26 0 > $O/r6_07_one_26.log 2>&1
db=$(find $O/hl -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/r6_07_headline_kernels.txt 2>&1; rm -rf $O/hl
head -30 $O/r6_07_headline_kernels.txt | cut -c1-160
timeout 600 python $R/tools/gpu_msm_tail.py ab 12 16 20 23 26 > $O/r6_07_msm_sizes.log 2>&1; grep "auto" $O/r6_07_msm_sizes.log
