# round 4, job 13: the work-bound bucket-sum levels with the one-wave-per-SIMD paired-product kernels (SPPARK_MSM_LAT_LANES) against the two-wave ones
set -x
R=$PWD; mkdir -p $R/gpurun_out
for v in 65536 4000000000 65536 4000000000; do
  echo "SPPARK_MSM_LAT_LANES=$v" >> $R/gpurun_out/r4_13_lat_lanes.log
  timeout 300 env SPPARK_MSM_LAT_LANES=$v python tools/gpu_msm_tail.py sort 26 24 23 22 20 2>&1 | grep "auto" >> $R/gpurun_out/r4_13_lat_lanes.log
done
cat $R/gpurun_out/r4_13_lat_lanes.log
