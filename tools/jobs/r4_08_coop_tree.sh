# round 4, job 8: the record tree's narrow levels with cooperative additions; A/B against one wave per addition
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "tail_variants or msm_vs_oracle or golden or tunables or skewed or randomised or oversized or empty" > $R/gpurun_out/r4_08_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r4_08_pytest_msm.log
timeout 600 python tools/gpu_msm_tail.py ab 12 14 16 17 18 19 20 > $R/gpurun_out/r4_08_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_08_msm_sizes.log | grep -v "low-latency"
cd /tmp && export TMPDIR=/tmp; cd $R
for lg in 16; do
  rm -rf gpurun_out/prof_tl
  (cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r4_08_tl.log 2>&1)
  python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r4_08_msm_timeline_2p${lg}.txt 2>&1
  tail -22 gpurun_out/r4_08_msm_timeline_2p${lg}.txt | cut -c1-130
done
rm -rf gpurun_out/prof_tl
