set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest8.log 2>&1; tail -3 $R/gpurun_out/pytest8.log
timeout 300 python tools/gpu_msm_small_sweep.py 10 12 14 16 18 19 20 21 22 > $R/gpurun_out/r01_msm_small_sweep.log 2>&1; tail -3 $R/gpurun_out/r01_msm_small_sweep.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r01b $R/gpurun_out/prof_r01b_fetch $R/gpurun_out/prof_r01b_write
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01b -o bench -- python bench.py > $R/gpurun_out/bench8_prof.log 2>&1); tail -2 $R/gpurun_out/bench8_prof.log | cut -c1-600
(cd $R && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r01b_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/prof8_fetch.log 2>&1); tail -1 $R/gpurun_out/prof8_fetch.log | cut -c1-200
(cd $R && timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r01b_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ntt --no-extras > $R/gpurun_out/prof8_write.log 2>&1); tail -1 $R/gpurun_out/prof8_write.log | cut -c1-200
cd $R
timeout 300 python bench.py > $R/gpurun_out/bench8.log 2>&1; tail -1 $R/gpurun_out/bench8.log | cut -c1-1500
find gpurun_out/prof_r01b* -name "*.db" | head; du -sh gpurun_out
