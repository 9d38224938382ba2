# round 2, job 12: register/LDS-staged k_sortB, LDS-staged level-A scatter
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 200 python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_sort_new.log 2>&1; tail -1 $R/gpurun_out/r2_sort_new.log
SPPARK_EXP_DIRECT_SCATTER=1 timeout 200 python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_sort_new_direct.log 2>&1; tail -1 $R/gpurun_out/r2_sort_new_direct.log
timeout 1200 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest12.log 2>&1; tail -3 $R/gpurun_out/r2_pytest12.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_tl.log 2>&1)
cd $R
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 36 > gpurun_out/r2_msm_timeline2.txt 2>&1
grep -v "reduce_runs\|levelN" gpurun_out/r2_msm_timeline2.txt | tail -22
rm -rf gpurun_out/prof_tl
