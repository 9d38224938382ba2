# round 2, job 13: instruction-rate microbenchmark (carry ops) + k_sortB with the cross-lane block scan
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 120 tools/exp/ubench_carry > $R/gpurun_out/r2_ubench_carry.log 2>&1; cat $R/gpurun_out/r2_ubench_carry.log
timeout 200 python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_sort_new2.log 2>&1; tail -1 $R/gpurun_out/r2_sort_new2.log
timeout 1200 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest13.log 2>&1; tail -3 $R/gpurun_out/r2_pytest13.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_tl.log 2>&1)
cd $R
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 36 > gpurun_out/r2_msm_timeline3.txt 2>&1
grep "sortB\|scatterA\|accumulate\|span" gpurun_out/r2_msm_timeline3.txt
rm -rf gpurun_out/prof_tl
