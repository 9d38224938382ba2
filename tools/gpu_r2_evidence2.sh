# round 2 evidence run (second half of the round): full GPU suite, rocprofv3 kernel trace + stats of
# bench.py, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate, no trace domains), then bench.py itself.
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r2_pytest_final.log 2>&1; tail -3 $R/gpurun_out/r2_pytest_final.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02c $R/gpurun_out/prof_r02c_fetch $R/gpurun_out/prof_r02c_write
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02c -o bench -- python bench.py > $R/gpurun_out/r2c_bench_prof.log 2>&1); tail -c 300 $R/gpurun_out/r2c_bench_prof.log
(cd $R && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r02c_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r2c_prof_fetch.log 2>&1); tail -c 200 $R/gpurun_out/r2c_prof_fetch.log
(cd $R && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r02c_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r2c_prof_write.log 2>&1); tail -c 200 $R/gpurun_out/r2c_prof_write.log
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r02c -name "*.db" | head -1) $(find gpurun_out/prof_r02c_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r02c_write -name "*.db" | head -1) > gpurun_out/r02c_bench_rocprofv3_summary.txt 2>&1
python tools/rocprof_timeline.py $(find gpurun_out/prof_r02c -name "*.db" | head -1) 400 > gpurun_out/r02c_timeline_all.txt 2>&1
head -30 gpurun_out/r02c_bench_rocprofv3_summary.txt | cut -c1-140
timeout 600 python bench.py > $R/gpurun_out/r2c_bench_final.json 2> $R/gpurun_out/r2c_bench_final.err; tail -c 400 $R/gpurun_out/r2c_bench_final.json
timeout 100 tools/exp/ubench_carry > $R/gpurun_out/r02_ubench_instruction_rates.log 2>&1
rm -rf gpurun_out/prof_r02c gpurun_out/prof_r02c_fetch gpurun_out/prof_r02c_write
du -sh gpurun_out
