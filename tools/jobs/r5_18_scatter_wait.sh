R=$PWD; mkdir -p $R/gpurun_out; out=$R/gpurun_out/r5_18
timeout 1500 python -m pytest tests/test_msm_gpu.py -x -q -m gpu > $out.tests.log 2>&1; grep -n "passed\|failed\|Error" $out.tests.log | head -5
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5w
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5w -o msm -- python tools/gpu_msm_records_one.py packed > /dev/null 2>&1)
(cd $R && python tools/rocprof_summary.py $(find gpurun_out/prof_r5w -name "*.db" | head -1) 2>&1 | grep -i "kernel \|scatterA\|sortB\|histA\|breakdown\|convert\|accumulate" | head -10 | cut -c1-130 | tee $out.kernels.log)
rm -rf $R/gpurun_out/prof_r5w
cd $R; timeout 300 python tools/gpu_msm_records_ab.py 26 24 20 2>&1 | grep "^2\^" | tee -a $out.kernels.log
