# round 2 evidence run: rocprofv3 kernel trace + stats of bench.py, the two PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate, no trace domains), SQ counters of the NTT pass, then bench.py itself.
set -x
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02 $R/gpurun_out/prof_r02_fetch $R/gpurun_out/prof_r02_write $R/gpurun_out/prof_r02_ntt*
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02 -o bench -- python bench.py > $R/gpurun_out/r2_bench_prof.log 2>&1); tail -c 400 $R/gpurun_out/r2_bench_prof.log
(cd $R && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r02_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r2_prof_fetch.log 2>&1); tail -c 200 $R/gpurun_out/r2_prof_fetch.log
(cd $R && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r02_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r2_prof_write.log 2>&1); tail -c 200 $R/gpurun_out/r2_prof_write.log
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  (cd $R && timeout 200 rocprofv3 --pmc $set -d $R/gpurun_out/prof_r02_ntt_$tag -o ntt -- python tools/gpu_ntt_one.py gl64 24 6 > $R/gpurun_out/r2_pmc_ntt_$tag.log 2>&1)
done
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r02 -name "*.db" | head -1) $(find gpurun_out/prof_r02_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r02_write -name "*.db" | head -1) > gpurun_out/r02_bench_rocprofv3_summary.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_r02 -name "*.db" | head -1) $(find gpurun_out/prof_r02_ntt_SQ_WAVES -name "*.db" | head -1) $(find gpurun_out/prof_r02_ntt_SQ_ACTIVE_INST_VALU -name "*.db" | head -1) | sed -n '/--pmc/,$p' > gpurun_out/r02_ntt_gl64_pmc.txt 2>&1
head -40 gpurun_out/r02_bench_rocprofv3_summary.txt
timeout 600 python bench.py > $R/gpurun_out/r2_bench_final.json 2> $R/gpurun_out/r2_bench_final.err; tail -c 600 $R/gpurun_out/r2_bench_final.json
# keep the databases out of the merge (size): only summaries travel back
rm -rf gpurun_out/prof_r02 gpurun_out/prof_r02_fetch gpurun_out/prof_r02_write gpurun_out/prof_r02_ntt_*
du -sh gpurun_out
