# k_piece_tail_coop with the adopted threshold (levels of <= 2^15 work items): MSM GPU tests, small-size walls.  Outputs: gpurun_out/r6_37_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests/test_msm_gpu.py tests/test_abi.py -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_37_pytest_msm.log 2>&1; tail -3 $R/gpurun_out/r6_37_pytest_msm.log
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=" > $R/gpurun_out/r6_37_small_wall.log; cat $R/gpurun_out/r6_37_small_wall.log
timeout 300 python tools/gpu_msm_tail.py ab 10 12 14 15 16 2>&1 | grep -v amdgpu | grep "auto" | tee -a $R/gpurun_out/r6_37_small_wall.log
for c in bn254 bls12_377 pallas; do timeout 200 python tools/gpu_msm_piece_tail.py $c 12 14 16 2>&1 | grep -v amdgpu | grep "auto\|per level" | sed "s/^/$c /" | tee -a $R/gpurun_out/r6_37_other_curves.log; done
