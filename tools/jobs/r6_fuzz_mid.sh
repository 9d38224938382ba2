# Randomised parity at the medium sizes over all-distinct points (tools/gpu_fuzz.py ... mid): two seeds.  Outputs: gpurun_out/r6_fuzz_mid.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6_fuzz_mid.log
for seed in 611 612; do
  timeout 400 python tools/gpu_fuzz.py 210 $seed mid 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_fuzz_mid.log
done
tail -20 $R/gpurun_out/r6_fuzz_mid.log
