#!/bin/bash
# Round 5, job 10: the G2 accumulation by wave pairs under the SQ counters (instructions per launch) and its kernel time per DISPATCH
# from a rocprofv3 kernel trace (the first call of a process touches fresh scratch and is slower): BLS12-381 G2 2^22, five calls.
R=$PWD; mkdir -p $R/gpurun_out
rm -f gpurun_out/pmc_g2_acc5.txt
bash tools/gpu_pmc_job.sh g2_acc5 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" -- python tools/gpu_g2_one.py bls12_381 22 5 | grep -i "accumulate\|kernel "
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_g2
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g2 -o g2 -- python tools/gpu_g2_one.py bls12_381 22 5 > $R/gpurun_out/r5_10_g2.log 2>&1)
cd $R; python tools/rocprof_summary.py $(find gpurun_out/prof_g2 -name "*.db" | head -1) > gpurun_out/r5_10_g2_rocprofv3_summary.txt 2>&1; head -8 gpurun_out/r5_10_g2_rocprofv3_summary.txt | cut -c1-130
python tools/rocprof_dispatches.py $(find gpurun_out/prof_g2 -name "*.db" | head -1) 2>&1 | grep -i "accumulate_g2c" | tee gpurun_out/r5_10_g2_dispatches.txt
rm -rf gpurun_out/prof_g2
