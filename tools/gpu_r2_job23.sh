set -x
for v in A B A B; do cp tools/exp/lib${v}_bn254.so sppark_amd/lib/libsppark_bn254.so; echo "variant $v $(timeout 200 python tools/gpu_msm_bn254.py 26 2>&1 | tail -1)"; done
cp tools/exp/libB_bn254.so sppark_amd/lib/libsppark_bn254.so
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "bn254" 2>&1 | tail -2
