# round 2, job 21: A = product build; C = lean madd order, single chains, no prefetch, 3 waves (16 spills); D = the same at 2 waves; E = lean order with prefetch, 2 waves
set -x
R=$PWD
mkdir -p $R/gpurun_out
for v in A C D E A C; do cp tools/exp/lib${v}_bls12_381.so sppark_amd/lib/libsppark_bls12_381.so; echo "variant $v $(timeout 200 python tools/gpu_msm_one.py 26 0 2>&1 | tail -1)"; done > $R/gpurun_out/r2_acc_lean_ab.log; cat $R/gpurun_out/r2_acc_lean_ab.log
cp tools/exp/libA_bls12_381.so sppark_amd/lib/libsppark_bls12_381.so
