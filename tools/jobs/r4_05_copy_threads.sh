# round 4, job 5: the points copy of a host chunk in T slices from T host threads (SPPARK_MSM_COPY_THREADS)
set -x
R=$PWD; mkdir -p $R/gpurun_out
for T in 1 2 4 1 2 4; do
  echo "SPPARK_MSM_COPY_THREADS=$T" >> $R/gpurun_out/r4_05_copy_threads.log
  timeout 300 env SPPARK_MSM_COPY_THREADS=$T python tools/gpu_msm_host.py 26 2>&1 | grep "chunk auto\|chunk 2^23\|chunk 2^24\|n/6" >> $R/gpurun_out/r4_05_copy_threads.log
done
cat $R/gpurun_out/r4_05_copy_threads.log
