# Timeline of the 2^20 and 2^18 MSMs (where the tail is a quarter of the call).  Outputs: gpurun_out/r6_tl_2p20.txt, r6_tl_2p18.txt
set -x
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lg in 20 18; do
  rm -rf $R/gpurun_out/prof_tl
  (cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r6_tl.log 2>&1)
  (cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 60 > gpurun_out/r6_tl_2p$lg.txt 2>&1)
done
rm -rf $R/gpurun_out/prof_tl
tail -45 $R/gpurun_out/r6_tl_2p20.txt | cut -c1-170
