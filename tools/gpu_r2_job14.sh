# round 2, job 14: Goldilocks pass kernel at 5 waves per SIMD (94 VGPRs)
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 300 python tools/gpu_ntt_bench.py > $R/gpurun_out/r2_ntt_w5.log 2>&1; grep -i "gl64\|goldilocks" $R/gpurun_out/r2_ntt_w5.log | head -20
timeout 900 python -m pytest tests/test_ntt_gpu.py -m gpu -x -q -k "gl64" > $R/gpurun_out/r2_pytest14.log 2>&1; tail -3 $R/gpurun_out/r2_pytest14.log
