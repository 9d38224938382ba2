# round 2, job 16: bucket-sum kernels at two waves per SIMD
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest16.log 2>&1; tail -3 $R/gpurun_out/r2_pytest16.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_tl.log 2>&1); tail -1 $R/gpurun_out/r2_tl.log
cd $R
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 36 > gpurun_out/r2_msm_timeline4.txt 2>&1
grep -v "big_\|scan_" gpurun_out/r2_msm_timeline4.txt | tail -34
rm -rf gpurun_out/prof_tl
timeout 200 python tools/gpu_msm_bn254.py 26 2>&1 | tail -1
timeout 200 python tools/gpu_g2_bench.py 2>&1 | tail -3
