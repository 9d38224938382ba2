# round 3, job 1: the radix-64 NTT plan (k_ntt6 / k_ntt12) -- parity on the GPU, then A/B against the 8-stage plan
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_ntt_gpu.py -m gpu -x -q -k "gl64 or bb31 or golden or small_field" > $R/gpurun_out/r3_01_pytest_ntt.log 2>&1; tail -5 $R/gpurun_out/r3_01_pytest_ntt.log
L=$R/gpurun_out/r3_01_ntt_ab.log; : > $L
for cfg in "SPPARK_NTT_R64_MIN=99" "SPPARK_NTT_R64_MIN=12" "SPPARK_NTT_R64_MIN=12 SPPARK_NTT_R64_DIRECT=24" "SPPARK_NTT_R64_MIN=12 SPPARK_NTT_R64_DIRECT=12"; do
  echo "=== $cfg" >> $L
  env $cfg NTT_FIELDS=gl64,bb31 NTT_LGS=12,14,16,18,20,22,24,26 timeout 300 python tools/gpu_ntt_bench.py >> $L 2>&1
done
cat $L | grep -v amdgpu.ids
