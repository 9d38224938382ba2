#!/bin/bash
# round 4, job 27: what a small sppark_ntt call costs on the host (tools/exp/host_overhead.hip)
mkdir -p gpurun_out
./tools/exp/host_overhead sppark_amd/lib/libsppark_gl64.so sppark_amd/lib/libsppark_bls12_381.so 2>&1 | tee gpurun_out/r4_27_host_overhead.log
