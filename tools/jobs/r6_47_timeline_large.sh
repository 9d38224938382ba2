# Kernel timelines of 2^23 / 2^24 / 2^25 on the final tree (their tails per bucket are twice those of 2^25 / 2^26).  Outputs: gpurun_out/r6_47_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lg in 23 24 25; do
  rm -rf $R/gpurun_out/prof_tl
  (cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r6_47_tl.log 2>&1)
  (cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 40 > gpurun_out/r6_47_msm_timeline_2p$lg.txt 2>&1)
done
rm -rf $R/gpurun_out/prof_tl
