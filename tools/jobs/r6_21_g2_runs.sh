# G2 run lengths of the wave-pair accumulation adopted in the plan: G2 parity tests, the G2 sizes.  Outputs: gpurun_out/r6_21_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "g2" > $R/gpurun_out/r6_21_pytest_g2.log 2>&1; tail -3 $R/gpurun_out/r6_21_pytest_g2.log
timeout 300 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_21_msm_g2.log; cat $R/gpurun_out/r6_21_msm_g2.log
