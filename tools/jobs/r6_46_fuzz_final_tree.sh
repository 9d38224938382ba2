# Randomised parity on the final tree of the round (LDE hand-overs, two-wave first level): small, medium, API.  Outputs: gpurun_out/r6_46_fuzz.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6_46_fuzz.log
for seed in 921 922; do timeout 400 python tools/gpu_fuzz.py 240 $seed 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_46_fuzz.log; done
for seed in 931 932; do timeout 400 python tools/gpu_fuzz.py 200 $seed mid 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_46_fuzz.log; done
for seed in 941; do timeout 400 python tools/gpu_fuzz.py 200 $seed api 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_46_fuzz.log; done
cat $R/gpurun_out/r6_46_fuzz.log | cut -c1-300
