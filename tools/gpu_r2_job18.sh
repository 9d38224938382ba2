# round 2, job 18: subset-sum top of the bucket sums
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest18.log 2>&1; tail -3 $R/gpurun_out/r2_pytest18.log
for top in 1 512 4096 32768; do for lg in 16 20 23 26; do echo "top $top lg $lg $(SPPARK_TOP=$top timeout 200 python tools/gpu_msm_one.py $lg 0 2>&1 | tail -1)"; done; done > $R/gpurun_out/r2_top_sweep.log; cat $R/gpurun_out/r2_top_sweep.log
