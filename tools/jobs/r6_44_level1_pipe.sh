# The first bucket-sum level's two chains on two waves (k_bucket_level1_pipe): MSM tests, A/B over the medium sizes and curves, a short
# randomised run over the medium sizes.  Outputs: gpurun_out/r6_44_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_44_pytest_msm.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_44_pytest_msm.log
timeout 600 python tools/gpu_msm_level1_ab.py 16 17 18 19 20 21 22 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_44_level1_ab.log
for c in bn254 bls12_377 pallas; do timeout 300 python tools/gpu_msm_level1_ab.py $c 18 20 2>&1 | grep -v amdgpu | tee -a $R/gpurun_out/r6_44_level1_ab.log; done
timeout 300 python tools/gpu_fuzz.py 150 911 mid 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_44_fuzz_mid.log | cut -c1-300
