#!/bin/bash
# round 4, job 24: the two NTTs timed beside each other on one box (tools/gpu_ntt_vs_reference.py)
mkdir -p gpurun_out
timeout 600 python tools/gpu_ntt_vs_reference.py 2>&1 | tee gpurun_out/r4_24_ntt_vs_reference_timing.log | tail -40
