# round 2, job 20: A/B on one box: k_accumulate at 2 waves per SIMD (A), 3 waves without the point prefetch (C),
# 3 waves with it (D: 93 spilled registers), 2 waves without prefetch (E)
set -x
R=$PWD
mkdir -p $R/gpurun_out
for v in A C D E A C; do cp tools/exp/lib${v}_bls12_381.so sppark_amd/lib/libsppark_bls12_381.so; echo "variant $v $(timeout 200 python tools/gpu_msm_one.py 26 0 2>&1 | tail -1)"; done > $R/gpurun_out/r2_acc_waves_ab.log; cat $R/gpurun_out/r2_acc_waves_ab.log
cp tools/exp/libA_bls12_381.so sppark_amd/lib/libsppark_bls12_381.so
