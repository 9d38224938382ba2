#!/bin/bash
# Round 6, job 4: coset transforms folded into the radix-64 passes: the NTT GPU suites (oracle + the reference's own build),
# then the A/B on one box (tuning build: SPPARK_NTT_COSET_FOLD=0 is the separate scaling launch).
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_ntt_gpu.py $R/tests/test_ntt_vs_reference_gpu.py -x -q --timeout 600 > $O/r6_04_pytest.log 2>&1; tail -3 $O/r6_04_pytest.log
{
for fold in 1 0 1 0; do
  SPPARK_LIBDIR=lib_tuning SPPARK_NTT_COSET_FOLD=$fold NTT_LGS=18,24 timeout 300 python $R/tools/gpu_ntt_orders.py
done
echo "== shipped libraries"
NTT_LGS=12,18,20,24 timeout 300 python $R/tools/gpu_ntt_orders.py
} > $O/r6_04_ntt_orders_ab.log 2>&1
grep "2^24" $O/r6_04_ntt_orders_ab.log | cut -c1-200
