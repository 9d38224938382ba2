# round 3, job 2: k_join_runs (short record segments in one launch) -- full GPU suite, tail sweep, dispatch timelines
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r3_02_pytest_gpu.log 2>&1; tail -5 $R/gpurun_out/r3_02_pytest_gpu.log
timeout 900 python tools/gpu_msm_tail.py 16 20 23 26 > $R/gpurun_out/r3_02_msm_tail.log 2>&1; grep -v amdgpu.ids $R/gpurun_out/r3_02_msm_tail.log
cd /tmp && export TMPDIR=/tmp
for lg in 23 20 16; do
  rm -rf $R/gpurun_out/prof_tl
  (cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r3_02_tl.log 2>&1); tail -2 $R/gpurun_out/r3_02_tl.log
  (cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r3_02_msm_timeline_2p$lg.txt 2>&1)
  cat $R/gpurun_out/r3_02_msm_timeline_2p$lg.txt
done
rm -rf $R/gpurun_out/prof_tl
