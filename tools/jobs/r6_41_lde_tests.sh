# The NTT / LDE / polynomial GPU suites on the tree with sppark_lde's hand-overs inside the transforms' steps.  Outputs: gpurun_out/r6_41_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_ntt_gpu.py tests/test_ntt_vs_reference_gpu.py tests/test_poly_gpu.py -m gpu -q --timeout 600 > $R/gpurun_out/r6_41_pytest_ntt.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_41_pytest_ntt.log
timeout 300 python tools/gpu_fuzz.py 120 901 api 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_41_fuzz_api.log | cut -c1-300
