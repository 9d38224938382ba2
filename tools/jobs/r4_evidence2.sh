# round 4, second evidence run (after the NTT changes: the reference's HIP build as oracle / baseline, the 256-bit
# fields' one-stage-per-round passes, device-resident NTT timings on a non-null stream): full GPU suite, rocprofv3
# kernel trace + stats of bench.py, bench.py itself, the NTT tables.  The MSM kernels are those of r4_evidence.sh.
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 300 > $R/gpurun_out/r4f_pytest_gpu.log 2>&1; tail -3 $R/gpurun_out/r4f_pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r4f
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4f -o bench -- python bench.py > $R/gpurun_out/r4f_bench_prof.log 2>&1); tail -c 300 $R/gpurun_out/r4f_bench_prof.log
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r4f -name "*.db" | head -1) > gpurun_out/r4f_bench_rocprofv3_summary.txt 2>&1
head -30 gpurun_out/r4f_bench_rocprofv3_summary.txt | cut -c1-140
timeout 900 python bench.py > $R/gpurun_out/r4f_bench_final.json 2> $R/gpurun_out/r4f_bench_final.err; tail -c 900 $R/gpurun_out/r4f_bench_final.json
timeout 400 env NTT_LGS=12,16,20,22,24,26 python tools/gpu_ntt_bench.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r4f_ntt_bench.log; cat $R/gpurun_out/r4f_ntt_bench.log | cut -c1-220
for spec in "gl64 22 2" "gl64 20 3" "bb31 22 2" "bls12_381 20 2"; do timeout 120 python tools/gpu_lde_one.py $spec 2>&1 | grep LDE >> $R/gpurun_out/r4f_ntt_lde.log; done; cat $R/gpurun_out/r4f_ntt_lde.log
rm -rf gpurun_out/prof_r4f
