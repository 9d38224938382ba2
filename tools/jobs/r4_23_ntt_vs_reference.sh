#!/bin/bash
# round 4, job 23: sppark_amd's NTT against the reference's own NTT (its HIP path, built for gfx950 into oracle/_ref by
# `make -C oracle ref_ntt`) on the same MI355X: parity tests, then both timed on device-resident 2^24 / 2^22 arrays.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ntt_vs_reference_gpu.py -q -m gpu --timeout 400 --durations=8 2>&1 | tee gpurun_out/r4_23_ntt_vs_reference.log | tail -40
