# Round 5 evidence run on ONE box with the final tree: full GPU suite; rocprofv3 kernel trace + stats of bench.py; the PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate, counters only) of the MSM and of the Goldilocks NTT -> the two traffic JSONs bench.py
# reads; SQ counters of k_accumulate (BLS12-381 and alt_bn128); the 2^26 timeline; then bench.py itself and the side tables.
# Every command under its own timeout.  Outputs: gpurun_out/r5e_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $R/gpurun_out/r5e_pytest_gpu.log 2>&1; tail -3 $R/gpurun_out/r5e_pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5e $R/gpurun_out/prof_r5e_fetch $R/gpurun_out/prof_r5e_write
(cd $R && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r5e_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r5e_prof_fetch.log 2>&1); tail -c 200 $R/gpurun_out/r5e_prof_fetch.log
(cd $R && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r5e_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r5e_prof_write.log 2>&1); tail -c 200 $R/gpurun_out/r5e_prof_write.log
cd $R
python tools/make_pmc_traffic.py $(find gpurun_out/prof_r5e_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r5e_write -name "*.db" | head -1) 26 > gpurun_out/r5e_pmc_traffic.json 2>&1
head -12 gpurun_out/r5e_pmc_traffic.json
# Goldilocks NTT 2^24 forward NR: FETCH / WRITE passes of six transforms
rm -rf gpurun_out/prof_r5e_nf gpurun_out/prof_r5e_nw
(cd /tmp && cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r5e_nf -o ntt -- python tools/gpu_ntt_one.py gl64 24 6 > $R/gpurun_out/r5e_prof_nf.log 2>&1)
(cd /tmp && cd $R && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r5e_nw -o ntt -- python tools/gpu_ntt_one.py gl64 24 6 > $R/gpurun_out/r5e_prof_nw.log 2>&1)
python tools/make_ntt_pmc_traffic.py $(find gpurun_out/prof_r5e_nf -name "*.db" | head -1) $(find gpurun_out/prof_r5e_nw -name "*.db" | head -1) 24 6 > gpurun_out/r5e_ntt_gl64_pmc.json 2>&1
cat gpurun_out/r5e_ntt_gl64_pmc.json | head -30
# the two JSONs are what bench.py's traffic fields read: in place for the runs below (committed from gpurun_out afterwards)
python -c "import json,sys; json.load(open('gpurun_out/r5e_pmc_traffic.json')); json.load(open('gpurun_out/r5e_ntt_gl64_pmc.json'))" && cp gpurun_out/r5e_pmc_traffic.json profiles/r05_pmc_traffic.json && cp gpurun_out/r5e_ntt_gl64_pmc.json profiles/r05_ntt_gl64_pmc.json
(cd /tmp && cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5e -o bench -- python bench.py > $R/gpurun_out/r5e_bench_prof.log 2>&1); tail -c 300 $R/gpurun_out/r5e_bench_prof.log
python tools/rocprof_summary.py $(find gpurun_out/prof_r5e -name "*.db" | head -1) $(find gpurun_out/prof_r5e_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r5e_write -name "*.db" | head -1) > gpurun_out/r5e_bench_rocprofv3_summary.txt 2>&1
head -34 gpurun_out/r5e_bench_rocprofv3_summary.txt | cut -c1-140
rm -rf gpurun_out/prof_tl
(cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r5e_tl.log 2>&1)
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r5e_msm_timeline_2p26.txt 2>&1
rm -f gpurun_out/pmc_msm_acc5.txt gpurun_out/pmc_bn254_acc5.txt
bash tools/gpu_pmc_job.sh msm_acc5 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" -- python tools/gpu_msm_one.py 26 0 | grep -i "accumulate\|kernel "
bash tools/gpu_pmc_job.sh bn254_acc5 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" -- python tools/gpu_msm_bn254.py 26 | grep -i "accumulate\|kernel "
timeout 900 python bench.py > $R/gpurun_out/r5e_bench_final.json 2> $R/gpurun_out/r5e_bench_final.err; tail -c 600 $R/gpurun_out/r5e_bench_final.json
timeout 400 env NTT_LGS=12,16,20,22,24,26 python tools/gpu_ntt_bench.py > $R/gpurun_out/r5e_ntt_bench.log 2>&1
timeout 400 python tools/gpu_ntt_vs_reference.py > $R/gpurun_out/r5e_ntt_vs_reference.log 2>&1; grep -v amdgpu $R/gpurun_out/r5e_ntt_vs_reference.log | tail -30
timeout 400 python tools/gpu_msm_tail.py ab 10 12 14 16 17 18 19 20 21 22 23 24 25 26 > $R/gpurun_out/r5e_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r5e_msm_sizes.log
timeout 600 python tools/gpu_msm_fixed.py 22:20 24 26 > $R/gpurun_out/r5e_msm_fixed_base.log 2>&1; grep -v amdgpu $R/gpurun_out/r5e_msm_fixed_base.log
for lg in 26 24 22 20 16; do timeout 200 python tools/gpu_msm_bn254.py $lg 2>&1 | grep -v amdgpu | tail -1 >> $R/gpurun_out/r5e_msm_bn254.log; done; cat $R/gpurun_out/r5e_msm_bn254.log
for spec in "gl64 22 2" "gl64 20 3" "bb31 22 2" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> $R/gpurun_out/r5e_ntt_lde.log; done; timeout 120 python tools/gpu_poly_one.py 2>&1 | grep -v amdgpu >> $R/gpurun_out/r5e_ntt_lde.log; cat $R/gpurun_out/r5e_ntt_lde.log
timeout 200 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r5e_msm_g2.log; cat $R/gpurun_out/r5e_msm_g2.log
: > $R/gpurun_out/r5e_ntt_small.log
for o in 1 0 2 3; do echo "== order $o (0 NN, 1 NR, 2 RN, 3 RR)" >> $R/gpurun_out/r5e_ntt_small.log; timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep "^gl64\|^bb31\|^bls12_381\|^bn254\|rows" | grep "2^8 \|2^9 \|2^10 \|2^11 \|2^12 \|2^16 \|2^20 \|rows" >> $R/gpurun_out/r5e_ntt_small.log; done; grep "rows" $R/gpurun_out/r5e_ntt_small.log
timeout 300 python tools/gpu_msm_records_ab.py 26 24 22 2>&1 | grep "^2\^" > $R/gpurun_out/r5e_sort_records_ab.log; cat $R/gpurun_out/r5e_sort_records_ab.log
rm -rf gpurun_out/prof_r5e gpurun_out/prof_r5e_fetch gpurun_out/prof_r5e_write gpurun_out/prof_tl gpurun_out/prof_r5e_nf gpurun_out/prof_r5e_nw
du -sh gpurun_out
