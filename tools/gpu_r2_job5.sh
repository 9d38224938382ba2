# round 2, job 5: fused Y3 reduction (MSM), Goldilocks shift folds + pass twiddle tables (NTT)
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest5.log 2>&1; tail -5 $R/gpurun_out/r2_pytest5.log
timeout 300 python tools/gpu_ntt_bench.py > $R/gpurun_out/r2_ntt_bench5.log 2>&1; cat $R/gpurun_out/r2_ntt_bench5.log
timeout 300 python tools/gpu_msm_L.py 26 > $R/gpurun_out/r2_msm_L.log 2>&1; cat $R/gpurun_out/r2_msm_L.log
