# sppark_lde with the spread + coset shift inside the forward transform's first step (k_ntt12<.., LDE>): NTT / LDE tests, timings,
# a kernel trace of one size.  Outputs: gpurun_out/r6_40_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_40_pytest_ntt.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_40_pytest_ntt.log
: > $R/gpurun_out/r6_40_lde.log
for spec in "gl64 22 2" "gl64 22 1" "gl64 21 3" "gl64 20 3" "gl64 20 4" "gl64 16 2" "gl64 12 2" "bb31 22 2" "bb31 21 3" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> $R/gpurun_out/r6_40_lde.log; done; cat $R/gpurun_out/r6_40_lde.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_lde
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lde -o lde -- python tools/gpu_lde_one.py gl64 22 2 > /dev/null 2>&1)
cd $R; python tools/rocprof_summary.py $(find gpurun_out/prof_lde -name "*.db" | head -1) 2>&1 | head -14 | cut -c1-150 | tee $R/gpurun_out/r6_40_lde_kernels.txt
rm -rf $R/gpurun_out/prof_lde
