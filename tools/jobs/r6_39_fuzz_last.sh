# The randomised parity runs on the last tree of the round (k_piece_tail_coop, host Horner on mulx; the small-size fuzz now also draws
# tune_tail 8 / 16 + x).  Outputs: gpurun_out/r6_39_fuzz.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
: > $R/gpurun_out/r6_39_fuzz.log
for seed in 801 802 803; do timeout 400 python tools/gpu_fuzz.py 240 $seed 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_39_fuzz.log; done
for seed in 811; do timeout 400 python tools/gpu_fuzz.py 200 $seed mid 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_39_fuzz.log; done
for seed in 821; do timeout 400 python tools/gpu_fuzz.py 200 $seed api 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6_39_fuzz.log; done
cat $R/gpurun_out/r6_39_fuzz.log | cut -c1-300
