# Run length 128 at 2^20 / 2^21 adopted: the MSM GPU tests, then the sizes.  Outputs: gpurun_out/r6_26_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_26_pytest_msm.log 2>&1; tail -2 $R/gpurun_out/r6_26_pytest_msm.log
timeout 400 python tools/gpu_msm_tail.py ab 19 20 21 22 2>&1 | grep -v amdgpu | grep "auto" > $R/gpurun_out/r6_26_sizes.log
for c in bn254 bls12_377 pallas; do timeout 300 python tools/gpu_msm_tail.py $c ab 20 21 2>&1 | grep -v amdgpu | grep "auto" | sed "s/^/$c /" >> $R/gpurun_out/r6_26_sizes.log; done
cat $R/gpurun_out/r6_26_sizes.log
