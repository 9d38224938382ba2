#!/bin/bash
# ROUND 5, job 1b (prepared at the end of round 4; run first in round 5: profiles/r05_ntt_lat_tail_ab.log).  The switch lost at every size
# and the code behind it is deleted: this script documents what was measured, it no longer runs against the current tree.
# SPPARK_NTT_LAT_TAIL=2|3: the small-half stages of a one-stage-per-round pass of the 256-bit fields in registers, fused
# with the store / load (ntt_lat_tail_dif / ntt_lat_head_dit).  Parity with the switch on, then the A/B by size.
mkdir -p gpurun_out; out=gpurun_out/next_ntt_lat_tail_ab.log; : > $out
for r in 2 3; do
  SPPARK_NTT_LAT_TAIL=$r timeout 400 python -m pytest tests/test_ntt_vs_reference_gpu.py tests/test_ntt_gpu.py -q -x -m gpu -k "bls12_381 or bn254 or lde" --timeout 120 2>&1 | tail -3 | tee -a $out
done
for r in 0 2 3; do
  echo "== SPPARK_NTT_LAT_TAIL=$r" | tee -a $out
  SPPARK_NTT_LAT_TAIL=$r NTT_FIELDS=bls12_381 NTT_LGS=16,18,20,22,24 timeout 200 python tools/gpu_ntt_bench.py 2>&1 | grep "2^" | cut -c1-150 | tee -a $out
done
