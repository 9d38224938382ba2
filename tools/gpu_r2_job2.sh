# round 2, second GPU job: three-stream pipeline: MSM tests, group sweep, stream priorities, host path
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest2.log 2>&1; tail -5 $R/gpurun_out/r2_pytest2.log
timeout 300 python tools/gpu_msm_groups.py 26 1 2 3 4 6 12 > $R/gpurun_out/r2_groups26_b.log 2>&1; cat $R/gpurun_out/r2_groups26_b.log
for pr in 11 22 02 20; do SPPARK_MSM_STREAM_PRIO=$pr timeout 200 python tools/gpu_msm_groups.py 26 4 6 2>&1 | grep -v amdgpu.ids | sed "s/^/prio $pr: /" | tee -a $R/gpurun_out/r2_prio.log; done
timeout 200 python tools/gpu_msm_groups.py 22 1 2 4 6 > $R/gpurun_out/r2_groups22_b.log 2>&1; cat $R/gpurun_out/r2_groups22_b.log
timeout 200 python tools/gpu_msm_groups.py 20 1 2 4 > $R/gpurun_out/r2_groups20_b.log 2>&1; cat $R/gpurun_out/r2_groups20_b.log
timeout 300 python tools/gpu_msm_host.py > $R/gpurun_out/r2_host.log 2>&1; cat $R/gpurun_out/r2_host.log
