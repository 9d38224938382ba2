# round 2, job 11: 254/255-bit curves on the 28-bit-limb field; memset + conversion beside the sort
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "bn254 or pallas or vesta" > $R/gpurun_out/r2_pytest11.log 2>&1; tail -3 $R/gpurun_out/r2_pytest11.log
timeout 200 python tools/gpu_msm_bn254.py 26 > $R/gpurun_out/r2_bn254_montx.log 2>&1; cat $R/gpurun_out/r2_bn254_montx.log
timeout 200 python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_side0.log 2>&1; tail -1 $R/gpurun_out/r2_side0.log
SPPARK_EXP_SIDE=1 timeout 200 python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_side1.log 2>&1; tail -1 $R/gpurun_out/r2_side1.log
