# Host Horner with the adc-chain field: the field bench on the box's host, the small-size walls, the MSM tests (every result passes
# through the host Horner).  Outputs: gpurun_out/r6_35_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I sppark_amd/csrc tools/host_field_bench.cpp -o /tmp/host_field_bench && /tmp/host_field_bench | tee $R/gpurun_out/r6_35_host_field.log
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=" > $R/gpurun_out/r6_35_small_wall.log; cat $R/gpurun_out/r6_35_small_wall.log
timeout 300 python tools/gpu_msm_tail.py ab 10 12 14 16 20 26 2>&1 | grep -v amdgpu | grep "auto" | tee -a $R/gpurun_out/r6_35_small_wall.log
timeout 1500 python -m pytest tests/test_msm_gpu.py tests/test_abi.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_35_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r6_35_pytest_msm.log
