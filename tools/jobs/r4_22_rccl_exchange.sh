#!/bin/bash
# round 4, job 22: the native one-process-per-GPU exchange (sppark_msm_rccl): a real RCCL communicator of one rank, and
# four "ranks" over the RCCL test double; the one-process multi-device entry points beside them.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -q -x -m gpu -k "rccl or multi_device" --durations=5 2>&1 | tee gpurun_out/r4_22_rccl.log | tail -25
