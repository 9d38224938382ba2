# round 4, job 9: the chunked bucket-sum levels of small grids with cooperative operations; whole MSM test file; A/B at every size
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r4_09_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r4_09_pytest_msm.log
timeout 600 python tools/gpu_msm_tail.py ab 10 12 14 16 17 18 20 22 23 24 26 > $R/gpurun_out/r4_09_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_09_msm_sizes.log | grep -v "low-latency\|join off"
cd /tmp && export TMPDIR=/tmp; cd $R
for lg in 16 18 23; do
  rm -rf gpurun_out/prof_tl
  (cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r4_09_tl.log 2>&1)
  python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r4_09_msm_timeline_2p${lg}.txt 2>&1
  tail -9 gpurun_out/r4_09_msm_timeline_2p${lg}.txt | cut -c1-130
done
rm -rf gpurun_out/prof_tl
