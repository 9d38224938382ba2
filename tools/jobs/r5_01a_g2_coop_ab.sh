#!/bin/bash
# ROUND 5, job 1a (prepared at the end of round 4; run first in round 5 on the tree of commit ec79306 + a66c35d: profiles/r05_g2_coop_ab.log).
# SPPARK_G2_COOP no longer exists: the wave-pair kernel won and is the default over the 14-limb fields (sppark_msm_g2_path for tests).
# The G2 accumulation with one Fp2 component per wave (SPPARK_G2_COOP=1; ec/xyzz2_coop.hpp, msm/msm_g2c_kernels.hpp):
# parity of the G2 GPU tests with the switch on, then the A/B at 2^20 / 2^22.  Everything under short timeouts.
mkdir -p gpurun_out; out=gpurun_out/next_g2_coop_ab.log; : > $out
SPPARK_G2_COOP=1 timeout 400 python -m pytest tests/test_msm_gpu.py -q -x -m gpu -k "g2" --timeout 120 2>&1 | tail -4 | tee -a $out
for sw in 0 1; do
  echo "== SPPARK_G2_COOP=$sw" | tee -a $out
  SPPARK_G2_COOP=$sw timeout 200 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu | tee -a $out
done
