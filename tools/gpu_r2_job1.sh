# round 2, first GPU job: the whole -m gpu suite, the window-group sweep, the bench line
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $R/gpurun_out/r2_pytest1.log 2>&1; tail -25 $R/gpurun_out/r2_pytest1.log
timeout 400 python tools/gpu_msm_groups.py 26 > $R/gpurun_out/r2_groups26.log 2>&1; cat $R/gpurun_out/r2_groups26.log
timeout 200 python tools/gpu_msm_groups.py 22 1 2 4 > $R/gpurun_out/r2_groups22.log 2>&1; cat $R/gpurun_out/r2_groups22.log
timeout 600 python bench.py > $R/gpurun_out/r2_bench1.json 2> $R/gpurun_out/r2_bench1.err; tail -c 3000 $R/gpurun_out/r2_bench1.json; tail -5 $R/gpurun_out/r2_bench1.err
