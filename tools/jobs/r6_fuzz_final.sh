# The three randomised parity runs again on the final tree (new seeds).  Outputs: gpurun_out/r6_fuzz_final.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6_fuzz_final.log
for seed in 701 702; do timeout 400 python tools/gpu_fuzz.py 240 $seed 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_fuzz_final.log; done
for seed in 711 712; do timeout 400 python tools/gpu_fuzz.py 200 $seed mid 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_fuzz_final.log; done
for seed in 721; do timeout 400 python tools/gpu_fuzz.py 200 $seed api 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_fuzz_final.log; done
cat $R/gpurun_out/r6_fuzz_final.log | cut -c1-300
