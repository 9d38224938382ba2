# The later chunked bucket-sum levels on three waves (k_bucket_levelN_pipe): MSM tests, A/B against one lane per work item (tune_tail 10
# switches both pipelined levels off), a short randomised run over the medium sizes.  Outputs: gpurun_out/r6_48_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_48_pytest_msm.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_48_pytest_msm.log
timeout 600 python tools/gpu_msm_level1_ab.py 20 21 22 23 24 26 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_48_levelN_ab.log
for c in bn254; do timeout 300 python tools/gpu_msm_level1_ab.py $c 22 23 26 2>&1 | grep -v amdgpu | tee -a $R/gpurun_out/r6_48_levelN_ab.log; done
timeout 300 python tools/gpu_fuzz.py 150 951 mid 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_48_fuzz_mid.log | cut -c1-300
