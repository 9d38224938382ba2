# The narrow end of the piece tree in one launch (k_piece_tail_coop): A/B against a launch per level (join 8) and a sweep of
# the first fused level (join 16 + x: levels of <= 2^x work items), MSM tests.  Outputs: gpurun_out/r6_36_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q --timeout 600 -k "piece or small or tail or golden or ragged or skew" > $R/gpurun_out/r6_36_pytest.log 2>&1; tail -3 $R/gpurun_out/r6_36_pytest.log
timeout 600 python tools/gpu_msm_piece_tail.py 10 11 12 13 14 15 16 17 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r6_36_piece_tail_ab.log
