#!/bin/bash
# round 4, job 33: the chunk-plane LDS layout of k_ntt_pass_lat: shapes at the large sizes again, small sizes, LDS counters
mkdir -p gpurun_out; out=gpurun_out/r4_33_ntt_wide_lat_planes.log; : > $out
for cfg in "0 -1 -1" "8 2 10" "8 1 9" "8 3 11" "7 3 10" "6 4 10" "6 3 9" "8 -1 -1"; do
  set -- $cfg
  echo "== SPPARK_NTT_LAT_SMAX=$1 LGC=$2 LGTILE=$3" | tee -a $out
  env SPPARK_NTT_LAT_SMAX=$1 $( [ $2 -ge 0 ] && echo SPPARK_NTT_LAT_LGC=$2 SPPARK_NTT_LAT_LGTILE=$3 ) NTT_FIELDS=bls12_381 NTT_LGS=16,18,20,22,24 timeout 300 python tools/gpu_ntt_bench.py 2>&1 | grep "2^" | cut -c1-150 | tee -a $out
done
rm -f gpurun_out/pmc_wide_lat2.txt
SPPARK_NTT_LAT_SMAX=8 SPPARK_NTT_LAT_LGC=2 SPPARK_NTT_LAT_LGTILE=10 bash tools/gpu_pmc_job.sh wide_lat2 "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES|SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" -- python tools/gpu_ntt_one.py bls12_381 24 4 > /dev/null
grep k_ntt_pass gpurun_out/pmc_wide_lat2.txt | cut -c1-135 | tee -a $out
