# Randomised parity through the other entry points (tools/gpu_fuzz.py ... api): two seeds.  Outputs: gpurun_out/r6_fuzz_api.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6_fuzz_api.log
for seed in 621 622; do
  timeout 300 python tools/gpu_fuzz.py 150 $seed api 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_fuzz_api.log
done
tail -30 $R/gpurun_out/r6_fuzz_api.log | cut -c1-400
