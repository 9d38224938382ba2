# round 4, job 16: run length fitted to whole rounds of resident waves; 8-bit windows at 2^15; parity tests + every size
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "msm_vs_oracle or golden or tunables or randomised or full_size or pipeline_medium or large_linearity or fixed_base_tables" > $R/gpurun_out/r4_16_pytest.log 2>&1; tail -3 $R/gpurun_out/r4_16_pytest.log
timeout 600 python tools/gpu_msm_tail.py ab 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 > $R/gpurun_out/r4_16_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_16_msm_sizes.log | grep "auto "
timeout 300 python tools/gpu_msm_tail.py bn254 ab 14 16 17 18 20 22 23 26 2>&1 | grep "auto " | tee $R/gpurun_out/r4_16_msm_bn254.log
