# round 2, job 19: A/B on one box: column shift as alignbit + lshr (A) or one 64-bit shift (B)
set -x
R=$PWD
mkdir -p $R/gpurun_out
for v in A B A B; do cp tools/exp/lib${v}_bls12_381.so sppark_amd/lib/libsppark_bls12_381.so; echo "variant $v $(timeout 200 python tools/gpu_msm_one.py 26 0 2>&1 | tail -1)"; done > $R/gpurun_out/r2_shift64_ab.log; cat $R/gpurun_out/r2_shift64_ab.log
cp tools/exp/libB_bls12_381.so sppark_amd/lib/libsppark_bls12_381.so
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "bls12_381" 2>&1 | tail -2
