#!/bin/bash
# round 4, job 34: the NTT test files on the final build (oracle + the reference's build), then the timing table
mkdir -p gpurun_out; out=gpurun_out/r4_34_ntt_final.log; : > $out
timeout 1200 python -m pytest tests/test_ntt_vs_reference_gpu.py tests/test_ntt_gpu.py tests/test_poly_gpu.py -q -x -m gpu 2>&1 | tail -4 | tee -a $out
( timeout 600 python tools/gpu_ntt_vs_reference.py; timeout 300 python tools/gpu_ntt_small_vs_reference.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_34_ntt_vs_reference_timing.log
