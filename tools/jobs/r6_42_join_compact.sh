# k_join_runs with compacted jobs: MSM tests, A/B over sizes and curves.  Outputs: gpurun_out/r6_42_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_42_pytest_msm.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_42_pytest_msm.log
timeout 600 python tools/gpu_msm_join_ab.py 18 19 20 21 22 23 24 26 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_42_join_ab.log
timeout 300 python tools/gpu_msm_join_ab.py bn254 19 20 22 26 2>&1 | grep -v amdgpu | tee -a $R/gpurun_out/r6_42_join_ab.log
