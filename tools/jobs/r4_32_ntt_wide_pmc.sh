#!/bin/bash
# round 4, job 32: counters of the 256-bit passes at 2^24, register passes (LAT_SMAX=0) and one stage per round (8 / 2 / 10)
mkdir -p gpurun_out; rm -f gpurun_out/pmc_wide_reg.txt gpurun_out/pmc_wide_lat.txt
SETS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES|SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS|SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"
SPPARK_NTT_LAT_SMAX=0 bash tools/gpu_pmc_job.sh wide_reg "$SETS" -- python tools/gpu_ntt_one.py bls12_381 24 4 > /dev/null
SPPARK_NTT_LAT_SMAX=8 SPPARK_NTT_LAT_LGC=2 SPPARK_NTT_LAT_LGTILE=10 bash tools/gpu_pmc_job.sh wide_lat "$SETS" -- python tools/gpu_ntt_one.py bls12_381 24 4 > /dev/null
( echo "== register passes"; cat gpurun_out/pmc_wide_reg.txt; echo "== one stage per round (8 stages, 4 columns, 1024 elements)"; cat gpurun_out/pmc_wide_lat.txt ) | grep -v "^kernel" | cut -c1-130 | tee gpurun_out/r4_32_ntt_wide_pmc.log
