#!/bin/bash
# round 4, job 28: the NTT timings on a non-null stream (the NULL stream made every timed transform synchronous), ours
# against the reference's build, large and small sizes
mkdir -p gpurun_out
( timeout 600 python tools/gpu_ntt_vs_reference.py; timeout 300 python tools/gpu_ntt_small_vs_reference.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_28_ntt_streams.log
