# round 4, job 6: cooperative point operations (four waves per operation): device test, the subset-sum top through them, A/B
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "bit_exact_with_wire or bucket_sum_top or tail_variants or msm_vs_oracle or golden or tunables" > $R/gpurun_out/r4_06_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r4_06_pytest_msm.log
timeout 600 python tools/gpu_msm_tail.py ab 14 16 18 20 22 23 24 26 > $R/gpurun_out/r4_06_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_06_msm_sizes.log
cd /tmp && export TMPDIR=/tmp; cd $R
for lg in 18; do
  rm -rf gpurun_out/prof_tl
  (cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r4_06_tl.log 2>&1)
  python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r4_06_msm_timeline_2p${lg}.txt 2>&1
  tail -12 gpurun_out/r4_06_msm_timeline_2p${lg}.txt | cut -c1-130
done
rm -rf gpurun_out/prof_tl
