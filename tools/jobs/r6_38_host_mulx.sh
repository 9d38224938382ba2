# The host product on mulx / adcx / adox against the adc-chain C product on the box's host; small-size walls with it.  Outputs: gpurun_out/r6_38_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I sppark_amd/csrc tools/host_field_bench.cpp -o /tmp/hfb && (/tmp/hfb 200000 ab; /tmp/hfb 1000 ab | head -1) | tee $R/gpurun_out/r6_38_host_field.log
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -DSPPARK_HOST_NO_MULX -I sppark_amd/csrc tools/host_field_bench.cpp -o /tmp/hfb0 && /tmp/hfb0 1000 | sed "s/^/C only: /" | tee -a $R/gpurun_out/r6_38_host_field.log
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=False" > $R/gpurun_out/r6_38_small_wall.log; cat $R/gpurun_out/r6_38_small_wall.log
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 -k "golden or g2 or ragged or small" > $R/gpurun_out/r6_38_pytest.log 2>&1; grep -n "passed\|failed" $R/gpurun_out/r6_38_pytest.log
