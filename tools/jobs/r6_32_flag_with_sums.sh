# The piece tree's flag handed over with the window sums (no memset, no copy of its own): MSM GPU tests, small walls, a short fuzz, 2^12 timeline.  Outputs: gpurun_out/r6_32_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_32_pytest_msm.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_32_pytest_msm.log
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=" > $R/gpurun_out/r6_32_small_wall.log; cat $R/gpurun_out/r6_32_small_wall.log
timeout 300 python tools/gpu_fuzz.py 120 801 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_32_fuzz.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 12 0 > $R/gpurun_out/r6_tl.log 2>&1)
(cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 22 > gpurun_out/r6_32_tl_2p12.txt 2>&1)
rm -rf $R/gpurun_out/prof_tl
tail -22 $R/gpurun_out/r6_32_tl_2p12.txt | cut -c1-150
