# round 3, job 5: k_reduce_tail (the narrow end of the record tree in one launch), fan-in 4 only up to 2^18
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r3_05_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r3_05_pytest_msm.log
timeout 400 python tools/gpu_msm_tail.py ab 12 14 16 18 20 22 23 24 > $R/gpurun_out/r3_05_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_05_msm_sizes.log
timeout 300 python tools/gpu_msm_tail.py bn254 ab 16 20 23 26 >> $R/gpurun_out/r3_05_msm_sizes.log 2>&1; tail -8 $R/gpurun_out/r3_05_msm_sizes.log
timeout 200 python tools/gpu_msm_skew.py > $R/gpurun_out/r3_05_msm_skew.log 2>&1; grep -v amdgpu $R/gpurun_out/r3_05_msm_skew.log | tail -12
cd /tmp && export TMPDIR=/tmp
for lg in 16 20; do
  rm -rf $R/gpurun_out/prof_tl
  (cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r3_05_tl.log 2>&1)
  (cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 36 > gpurun_out/r3_05_msm_timeline_2p$lg.txt 2>&1)
  tail -22 $R/gpurun_out/r3_05_msm_timeline_2p$lg.txt | cut -c1-120
done
rm -rf $R/gpurun_out/prof_tl
