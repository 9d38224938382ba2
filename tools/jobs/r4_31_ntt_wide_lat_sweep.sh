#!/bin/bash
# round 4, job 31: shapes of the one-stage-per-round passes at the large sizes (stages per pass, columns per row, tile)
mkdir -p gpurun_out; out=gpurun_out/r4_31_ntt_wide_lat_sweep.log; : > $out
for cfg in "0 -1 -1" "8 2 10" "8 3 11" "7 3 10" "7 4 11" "6 4 10" "6 3 9" "6 4 11" "5 4 9" "5 4 10" "4 4 8" "4 4 9"; do
  set -- $cfg
  echo "== SPPARK_NTT_LAT_SMAX=$1 LGC=$2 LGTILE=$3" | tee -a $out
  env SPPARK_NTT_LAT_SMAX=$1 $( [ $2 -ge 0 ] && echo SPPARK_NTT_LAT_LGC=$2 SPPARK_NTT_LAT_LGTILE=$3 ) NTT_FIELDS=bls12_381 NTT_LGS=20,22,24 timeout 300 python tools/gpu_ntt_bench.py 2>&1 | grep "2^" | cut -c1-150 | tee -a $out
done
