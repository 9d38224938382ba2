# Kernel timelines of the medium sizes (2^18, 2^20) on the last tree.  Outputs: gpurun_out/r6_43_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lg in 18 20; do
  rm -rf $R/gpurun_out/prof_tl
  (cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r6_43_tl.log 2>&1)
  (cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 60 > gpurun_out/r6_43_msm_timeline_2p$lg.txt 2>&1)
done
rm -rf $R/gpurun_out/prof_tl
