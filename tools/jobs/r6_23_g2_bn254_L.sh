# alt_bn128 G2 (one lane per addition): run length x window bits at 2^16 .. 2^22 (tuning build).  Outputs: gpurun_out/r6_23_g2_bn254_L.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
SPPARK_LIBDIR=lib_tuning timeout 900 python tools/gpu_g2_L.py bn254 16 18 19 20 21 22 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_23_g2_bn254_L.log
cut -c1-110 $R/gpurun_out/r6_23_g2_bn254_L.log
