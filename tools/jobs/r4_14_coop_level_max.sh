# round 4, job 14: cooperative chunked levels up to 24576 / 32768 work items (1.5 / 2 rounds of work-groups) against 16384
set -x
R=$PWD; mkdir -p $R/gpurun_out
for v in 16384 24576 32768 16384 24576 32768; do
  echo "SPPARK_MSM_COOP_LEVEL_MAX=$v" >> $R/gpurun_out/r4_14_coop_level_max.log
  timeout 300 env SPPARK_MSM_COOP_LEVEL_MAX=$v python tools/gpu_msm_tail.py sort 26 25 24 23 22 21 2>&1 | grep "auto" >> $R/gpurun_out/r4_14_coop_level_max.log
done
cat $R/gpurun_out/r4_14_coop_level_max.log
