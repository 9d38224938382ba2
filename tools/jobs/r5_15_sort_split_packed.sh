R=$PWD; mkdir -p $R/gpurun_out; out=$R/gpurun_out/r5_15
cd /tmp && export TMPDIR=/tmp
: > $out.split.log
for lb in 9 10 11; do
  rm -rf $R/gpurun_out/prof_r5s
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5s -o msm -- python tools/gpu_msm_records_one.py packed $lb > /dev/null 2>&1)
  echo "== k_lo bits $lb (2^$((21-lb)) level-A partitions)" | tee -a $out.split.log
  (cd $R && python tools/rocprof_summary.py $(find gpurun_out/prof_r5s -name "*.db" | head -1) 2>&1 | grep -i "scatterA\|sortB\|histA\|big_\|accumulate" | head -8 | cut -c1-130 | tee -a $out.split.log)
  rm -rf $R/gpurun_out/prof_r5s
done
