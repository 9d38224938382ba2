# round 2, third GPU job: grouped sort+accumulate with a CU-masked sort stream
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest3.log 2>&1; tail -5 $R/gpurun_out/r2_pytest3.log
rm -f $R/gpurun_out/r2_groups_cus.log
for cus in 0 4 8 16 32 64; do SPPARK_MSM_AUX_CUS=$cus timeout 200 python tools/gpu_msm_groups.py 26 1 2 3 4 6 2>&1 | grep -v amdgpu.ids | sed "s/^/aux_cus $cus: /" | tee -a $R/gpurun_out/r2_groups_cus.log; done
for cus in 8 16; do SPPARK_MSM_AUX_CUS=$cus timeout 200 python tools/gpu_msm_groups.py 22 1 2 4 2>&1 | grep -v amdgpu.ids | sed "s/^/aux_cus $cus: /" | tee -a $R/gpurun_out/r2_groups_cus.log; done
