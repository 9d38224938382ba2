# Round 6 evidence run on ONE box with the final tree: full GPU suite; the PMC passes (FETCH_SIZE / WRITE_SIZE, separate, counters
# only) of the MSM and of the Goldilocks NTT -> the two traffic JSONs bench.py reads; rocprofv3 kernel trace + stats of the headline
# alone and of the whole bench.py; SQ counters of k_accumulate (BLS12-381 and alt_bn128); the 2^26 / 2^16 / 2^12 timelines; bench.py
# itself (timed: the driver's run must stay below ~35 s) and the side tables of DESIGN.md section 0.  Every command under its own timeout.
# Outputs: gpurun_out/r6e_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 600 > $R/gpurun_out/r6e_pytest_gpu.log 2>&1; grep -n "passed\|failed\|rror" $R/gpurun_out/r6e_pytest_gpu.log | head -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r6e $R/gpurun_out/prof_r6e_fetch $R/gpurun_out/prof_r6e_write $R/gpurun_out/prof_r6h
(cd $R && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r6e_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r6e_prof_fetch.log 2>&1); tail -c 200 $R/gpurun_out/r6e_prof_fetch.log
(cd $R && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r6e_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r6e_prof_write.log 2>&1); tail -c 200 $R/gpurun_out/r6e_prof_write.log
cd $R
python tools/make_pmc_traffic.py $(find gpurun_out/prof_r6e_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r6e_write -name "*.db" | head -1) 26 > gpurun_out/r6e_pmc_traffic.json 2>&1
head -12 gpurun_out/r6e_pmc_traffic.json
# Goldilocks NTT 2^24 forward NR: FETCH / WRITE passes of six transforms
rm -rf gpurun_out/prof_r6e_nf gpurun_out/prof_r6e_nw
(cd /tmp && cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r6e_nf -o ntt -- python tools/gpu_ntt_one.py gl64 24 6 > $R/gpurun_out/r6e_prof_nf.log 2>&1)
(cd /tmp && cd $R && timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r6e_nw -o ntt -- python tools/gpu_ntt_one.py gl64 24 6 > $R/gpurun_out/r6e_prof_nw.log 2>&1)
python tools/make_ntt_pmc_traffic.py $(find gpurun_out/prof_r6e_nf -name "*.db" | head -1) $(find gpurun_out/prof_r6e_nw -name "*.db" | head -1) 24 6 > gpurun_out/r6e_ntt_gl64_pmc.json 2>&1
cat gpurun_out/r6e_ntt_gl64_pmc.json | head -30
# the two JSONs are what bench.py's traffic fields read: in place for the runs below (committed from gpurun_out afterwards)
python -c "import json,sys; json.load(open('gpurun_out/r6e_pmc_traffic.json')); json.load(open('gpurun_out/r6e_ntt_gl64_pmc.json'))" && cp gpurun_out/r6e_pmc_traffic.json profiles/r06_pmc_traffic.json && cp gpurun_out/r6e_ntt_gl64_pmc.json profiles/r06_ntt_gl64_pmc.json
# the headline alone: every k_accumulate call of the summary is a 2^26-point launch (its average must agree with roofline.kernel_ms)
(cd /tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6h -o bench -- python bench.py --no-extras --no-ntt --no-cpu-baseline > $R/gpurun_out/r6e_bench_headline.json 2> $R/gpurun_out/r6e_bench_headline.err)
python tools/rocprof_summary.py $(find gpurun_out/prof_r6h -name "*.db" | head -1) > gpurun_out/r6e_bench_headline_rocprofv3_summary.txt 2>&1
head -24 gpurun_out/r6e_bench_headline_rocprofv3_summary.txt | cut -c1-130
(cd /tmp && cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6e -o bench -- python bench.py > $R/gpurun_out/r6e_bench_prof.log 2>&1); tail -c 300 $R/gpurun_out/r6e_bench_prof.log
python tools/rocprof_summary.py $(find gpurun_out/prof_r6e -name "*.db" | head -1) $(find gpurun_out/prof_r6e_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r6e_write -name "*.db" | head -1) > gpurun_out/r6e_bench_rocprofv3_summary.txt 2>&1
head -34 gpurun_out/r6e_bench_rocprofv3_summary.txt | cut -c1-140
for lg in 26 16 12; do
  rm -rf gpurun_out/prof_tl
  (cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_tail.py ab $lg > $R/gpurun_out/r6e_tl.log 2>&1)
  python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 200 > gpurun_out/r6e_msm_timeline_all_2p$lg.txt 2>&1
done
rm -rf gpurun_out/prof_tl
(cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r6e_tl.log 2>&1)
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r6e_msm_timeline_2p26.txt 2>&1
rm -f gpurun_out/pmc_msm_acc6.txt gpurun_out/pmc_bn254_acc6.txt
bash tools/gpu_pmc_job.sh msm_acc6 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" -- python tools/gpu_msm_one.py 26 0 | grep -i "accumulate\|convert\|kernel "
bash tools/gpu_pmc_job.sh bn254_acc6 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" -- python tools/gpu_msm_bn254.py 26 | grep -i "accumulate\|kernel "
SECONDS=0; timeout 900 python bench.py > $R/gpurun_out/r6e_bench_final.json 2> $R/gpurun_out/r6e_bench_final.err; echo "bench.py wall: $SECONDS s" | tee $R/gpurun_out/r6e_bench_wall.txt; tail -c 600 $R/gpurun_out/r6e_bench_final.json
timeout 400 env NTT_LGS=12,16,20,22,24,26 python tools/gpu_ntt_bench.py > $R/gpurun_out/r6e_ntt_bench.log 2>&1
timeout 400 env NTT_LGS=12,16,18,20,22,24 python tools/gpu_ntt_orders.py > $R/gpurun_out/r6e_ntt_orders.log 2>&1; grep "2^24" $R/gpurun_out/r6e_ntt_orders.log | cut -c1-220
timeout 400 python tools/gpu_ntt_vs_reference.py > $R/gpurun_out/r6e_ntt_vs_reference.log 2>&1; grep -v amdgpu $R/gpurun_out/r6e_ntt_vs_reference.log | tail -30
timeout 400 python tools/gpu_msm_tail.py ab 10 12 14 15 16 17 18 19 20 21 22 23 24 25 26 > $R/gpurun_out/r6e_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r6e_msm_sizes.log | grep "auto\|no piece"
timeout 300 python tools/gpu_msm_timing_overhead.py 2>&1 | grep "timing=" > $R/gpurun_out/r6e_msm_small_wall.log; cat $R/gpurun_out/r6e_msm_small_wall.log
timeout 600 python tools/gpu_msm_tail.py grid 13 14 > $R/gpurun_out/r6e_msm_small_grid_2p13_2p14.log 2>&1; grep "best\|auto" $R/gpurun_out/r6e_msm_small_grid_2p13_2p14.log
timeout 600 python tools/gpu_msm_fixed.py 22:20 24 26 > $R/gpurun_out/r6e_msm_fixed_base.log 2>&1; grep -v amdgpu $R/gpurun_out/r6e_msm_fixed_base.log
for lg in 26 24 22 20 16; do timeout 200 python tools/gpu_msm_bn254.py $lg 2>&1 | grep -v amdgpu | tail -1 >> $R/gpurun_out/r6e_msm_bn254.log; done; cat $R/gpurun_out/r6e_msm_bn254.log
for spec in "gl64 22 2" "gl64 20 3" "bb31 22 2" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> $R/gpurun_out/r6e_ntt_lde.log; done; timeout 120 python tools/gpu_poly_one.py 2>&1 | grep -v amdgpu >> $R/gpurun_out/r6e_ntt_lde.log; cat $R/gpurun_out/r6e_ntt_lde.log
timeout 200 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r6e_msm_g2.log; cat $R/gpurun_out/r6e_msm_g2.log
timeout 300 python tools/gpu_msm_curves.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r6e_msm_other_curves.log; cat $R/gpurun_out/r6e_msm_other_curves.log
rm -rf gpurun_out/prof_r6e gpurun_out/prof_r6e_fetch gpurun_out/prof_r6e_write gpurun_out/prof_tl gpurun_out/prof_r6e_nf gpurun_out/prof_r6e_nw gpurun_out/prof_r6h
du -sh gpurun_out
