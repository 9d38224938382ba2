# round 4, job 11: the whole GPU suite on the full rebuild (cooperative kernels in all five curve libraries, BabyBear min-trick), NTT timings
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 600 > $R/gpurun_out/r4_11_pytest_gpu.log 2>&1; tail -4 $R/gpurun_out/r4_11_pytest_gpu.log
timeout 300 env NTT_FIELDS=gl64,bb31 NTT_LGS=16,20,24,26 python tools/gpu_ntt_bench.py > $R/gpurun_out/r4_11_ntt_bench.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_11_ntt_bench.log
timeout 300 python tools/gpu_msm_tail.py bn254 ab 16 20 23 26 2>&1 | grep -v "amdgpu\|low-latency\|join off" | tee $R/gpurun_out/r4_11_msm_bn254.log
