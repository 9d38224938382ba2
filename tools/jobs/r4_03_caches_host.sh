# round 4, job 3: the cache fixes (NTT tests, the new cache test, MSM tests that touch reserve / fixed base / tail), the
# host-buffer chunk sweep on this box, 32 x 32 bit-reversal tiles for BabyBear
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_poly_gpu.py -m gpu -x -q > $R/gpurun_out/r4_03_pytest_ntt.log 2>&1; tail -3 $R/gpurun_out/r4_03_pytest_ntt.log
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "fixed_base or tail_variants or one_shot or preloaded or concurrent or multi_device or skew" > $R/gpurun_out/r4_03_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r4_03_pytest_msm.log
for tb in 6 5 6 5; do timeout 120 env SPPARK_NTT_BITREV_TB=$tb NTT_FIELDS=bb31 NTT_LGS=20,24,26 python tools/gpu_ntt_bench.py 2>&1 | grep bb31 | sed "s/^/TB=$tb /" >> $R/gpurun_out/r4_03_bitrev_tb.log; done; cat $R/gpurun_out/r4_03_bitrev_tb.log | cut -c1-260
timeout 600 python tools/gpu_msm_host.py 24 26 > $R/gpurun_out/r4_03_msm_host.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_03_msm_host.log
