# round 2, job 7: BLS12-377 (G1/G2 MSM, Fr NTT, polynomial ops), 1024-lane sort work-groups, full suite, bench
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > $R/gpurun_out/r2_pytest7.log 2>&1; tail -18 $R/gpurun_out/r2_pytest7.log
timeout 200 python tools/gpu_msm_groups.py 26 1 > $R/gpurun_out/r2_groups_final.log 2>&1; cat $R/gpurun_out/r2_groups_final.log
timeout 200 python tools/gpu_msm_377.py > $R/gpurun_out/r2_msm_377.log 2>&1; cat $R/gpurun_out/r2_msm_377.log
timeout 600 python bench.py > $R/gpurun_out/r2_bench7.json 2> $R/gpurun_out/r2_bench7.err; tail -c 400 $R/gpurun_out/r2_bench7.json; tail -3 $R/gpurun_out/r2_bench7.err
