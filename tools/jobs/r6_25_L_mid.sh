# Run length around the 64 -> 128 step of the plan.  Outputs: gpurun_out/r6_25_L_mid.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
(for c in bls12_381 bn254 bls12_377 pallas; do timeout 300 python tools/gpu_msm_L_mid.py $c 20 21 22; done) 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_25_L_mid.log
cat $R/gpurun_out/r6_25_L_mid.log
