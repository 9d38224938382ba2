# round 4, job 21: the last chunked bucket-sum level sized to hand the top exactly its capacity (4096 items per window) instead of fewer
set -x
R=$PWD; mkdir -p $R/gpurun_out
for v in 0 1 0 1; do
  echo "SPPARK_MSM_LAND_ON_TOP=$v" >> $R/gpurun_out/r4_21_land_on_top.log
  timeout 300 env SPPARK_MSM_LAND_ON_TOP=$v python tools/gpu_msm_tail.py sort 26 25 24 23 22 21 20 19 2>&1 | grep "auto" >> $R/gpurun_out/r4_21_land_on_top.log
done
cat $R/gpurun_out/r4_21_land_on_top.log
timeout 300 env SPPARK_MSM_LAND_ON_TOP=1 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "bucket_sum_top or full_size or pipeline_medium or large_linearity" 2>&1 | tail -2
