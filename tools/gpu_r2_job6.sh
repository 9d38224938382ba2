# round 2, job 6: 128-byte point records: tests, bench, FETCH/WRITE passes
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_msm_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest6.log 2>&1; tail -3 $R/gpurun_out/r2_pytest6.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02b*
(cd $R && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r02b_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r2b_prof_fetch.log 2>&1)
(cd $R && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r02b_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/r2b_prof_write.log 2>&1)
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02b -o bench -- python bench.py > $R/gpurun_out/r2b_bench_prof.log 2>&1)
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r02b -name "*.db" | head -1) $(find gpurun_out/prof_r02b_fetch -name "*.db" | head -1) $(find gpurun_out/prof_r02b_write -name "*.db" | head -1) > gpurun_out/r02b_bench_rocprofv3_summary.txt 2>&1
grep -n "k_accumulate<montx\|k_convert_points<montx_dev<bls12_381_fp_p, 28>, false\|k_breakdown<mont_dev<bls12_381" gpurun_out/r02b_bench_rocprofv3_summary.txt
timeout 600 python bench.py > $R/gpurun_out/r2b_bench.json 2> $R/gpurun_out/r2b_bench.err; tail -c 300 $R/gpurun_out/r2b_bench.json
rm -rf gpurun_out/prof_r02b*
