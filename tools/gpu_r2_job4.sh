# round 2, job 4: new tests (polynomial ops, batch addition, MSM pipeline shapes) + bench line
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_poly_gpu.py tests/test_msm_gpu.py -m gpu -x -q --durations=8 > $R/gpurun_out/r2_pytest4.log 2>&1; tail -22 $R/gpurun_out/r2_pytest4.log
timeout 600 python bench.py > $R/gpurun_out/r2_bench4.json 2> $R/gpurun_out/r2_bench4.err; tail -c 2500 $R/gpurun_out/r2_bench4.json; tail -5 $R/gpurun_out/r2_bench4.err
