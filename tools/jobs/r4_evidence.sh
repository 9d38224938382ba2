# round 4 evidence run: full GPU suite, rocprofv3 kernel trace + stats of bench.py, the two PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate, counters only), SQ counters of k_accumulate, then bench.py itself.  Outputs: gpurun_out/r4e_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 300 > $R/gpurun_out/r4e_pytest_gpu.log 2>&1; tail -3 $R/gpurun_out/r4e_pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r4e $R/gpurun_out/prof_r4e_fetch $R/gpurun_out/prof_r4e_write
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4e -o bench -- python bench.py > $R/gpurun_out/r4e_bench_prof.log 2>&1); tail -c 300 $R/gpurun_out/r4e_bench_prof.log
(cd $R && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r4e_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r4e_prof_fetch.log 2>&1); tail -c 200 $R/gpurun_out/r4e_prof_fetch.log
(cd $R && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r4e_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r4e_prof_write.log 2>&1); tail -c 200 $R/gpurun_out/r4e_prof_write.log
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r4e -name "*.db" | head -1) $(find gpurun_out/prof_r4e_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r4e_write -name "*.db" | head -1) > gpurun_out/r4e_bench_rocprofv3_summary.txt 2>&1
python tools/make_pmc_traffic.py $(find gpurun_out/prof_r4e_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r4e_write -name "*.db" | head -1) 26 > gpurun_out/r4e_pmc_traffic.json 2>&1
head -34 gpurun_out/r4e_bench_rocprofv3_summary.txt | cut -c1-140
cat gpurun_out/r4e_pmc_traffic.json | head -20
for lg in 26 23 20 18 16; do
  rm -rf gpurun_out/prof_tl
  (cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r4e_tl.log 2>&1)
  python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r4e_msm_timeline_2p$lg.txt 2>&1
done
rm -f gpurun_out/pmc_msm_acc3.txt
bash tools/gpu_pmc_job.sh msm_acc3 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" -- python tools/gpu_msm_one.py 26 0 | grep -i "accumulate\|kernel "
rm -f gpurun_out/pmc_ntt_bb31_r4e.txt gpurun_out/pmc_ntt_gl64_r4e.txt
bash tools/gpu_pmc_job.sh ntt_bb31_r4e "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|FETCH_SIZE|WRITE_SIZE" -- python tools/gpu_ntt_one.py bb31 24 6 | grep -i "k_ntt\|kernel "
bash tools/gpu_pmc_job.sh ntt_bb31_nn_r4e "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" -- python tools/gpu_ntt_one.py bb31 24 6 NN | grep -i "bitrev\|kernel "
timeout 900 python bench.py > $R/gpurun_out/r4e_bench_final.json 2> $R/gpurun_out/r4e_bench_final.err; tail -c 600 $R/gpurun_out/r4e_bench_final.json
timeout 400 env NTT_LGS=12,16,20,22,24,26 python tools/gpu_ntt_bench.py > $R/gpurun_out/r4e_ntt_bench.log 2>&1
timeout 400 python tools/gpu_msm_tail.py ab 10 12 14 16 17 18 19 20 21 22 23 24 25 26 > $R/gpurun_out/r4e_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4e_msm_sizes.log
timeout 600 python tools/gpu_msm_fixed.py 22:20 23 24 25 26 > $R/gpurun_out/r4e_msm_fixed_base.log 2>&1; timeout 300 python tools/gpu_msm_fixed.py bn254 24 26 >> $R/gpurun_out/r4e_msm_fixed_base.log 2>&1; grep -v amdgpu $R/gpurun_out/r4e_msm_fixed_base.log
for spec in "gl64 22 2" "gl64 20 3" "bb31 22 2" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> $R/gpurun_out/r4e_ntt_lde.log; done; timeout 120 python tools/gpu_poly_one.py 2>&1 | grep -v amdgpu >> $R/gpurun_out/r4e_ntt_lde.log; cat $R/gpurun_out/r4e_ntt_lde.log
timeout 200 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r4e_msm_g2.log; cat $R/gpurun_out/r4e_msm_g2.log
timeout 200 python tools/gpu_msm_skew.py > $R/gpurun_out/r4e_msm_skew.log 2>&1; grep -v amdgpu $R/gpurun_out/r4e_msm_skew.log | tail -8
rm -rf gpurun_out/prof_r4e gpurun_out/prof_r4e_fetch gpurun_out/prof_r4e_write gpurun_out/prof_tl
du -sh gpurun_out
