# round 4, job 4: half-sized first / last chunk of the host-buffer path; the cache test again
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_ntt_gpu.py -m gpu -x -q -k "cached_tables" > $R/gpurun_out/r4_04_pytest_ntt.log 2>&1; tail -3 $R/gpurun_out/r4_04_pytest_ntt.log
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "pipeline_medium or window_groups or one_shot or montgomery or multi_device" > $R/gpurun_out/r4_04_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r4_04_pytest_msm.log
timeout 600 python tools/gpu_msm_host.py 24 25 26 > $R/gpurun_out/r4_04_msm_host.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_04_msm_host.log
