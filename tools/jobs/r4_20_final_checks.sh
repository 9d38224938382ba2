# round 4, job 20: the pieces of the evidence set that the last plan change (8 % threshold of the run-length fit) touches, on the final build
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 400 python tools/gpu_msm_tail.py ab 15 16 17 18 19 20 21 22 23 > $R/gpurun_out/r4_20_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_20_msm_sizes.log | grep "auto \|no coop"
timeout 200 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r4_20_msm_g2.log; cat $R/gpurun_out/r4_20_msm_g2.log
timeout 200 python tools/gpu_msm_skew.py > $R/gpurun_out/r4_20_msm_skew.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_20_msm_skew.log | tail -8
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "g2 or randomised or tunables or msm_vs_oracle or skew or window_groups" > $R/gpurun_out/r4_20_pytest.log 2>&1; tail -2 $R/gpurun_out/r4_20_pytest.log
