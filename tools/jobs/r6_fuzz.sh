# Time-bounded randomised parity (tools/gpu_fuzz.py): three seeds, four minutes each.  Outputs: gpurun_out/r6_fuzz.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6_fuzz.log
for seed in 601 602 603; do
  timeout 400 python tools/gpu_fuzz.py 240 $seed 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_fuzz.log
done
cat $R/gpurun_out/r6_fuzz.log | tail -20
