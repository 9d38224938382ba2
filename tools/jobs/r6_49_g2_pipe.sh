# G2: the chunked bucket-sum levels on two / three waves (k_bucket_level1_pipe_g / _levelN_pipe_g): G2 tests (both accumulation kernels),
# timings (against profiles/r06_msm_g2.log of the evidence run before).  Outputs: gpurun_out/r6_49_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 -k "g2 or G2" > $R/gpurun_out/r6_49_pytest_g2.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_49_pytest_g2.log
timeout 600 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_49_g2_ab.log
