# round 2, job 8: Pasta curves (Pallas / Vesta MSM + NTT + polynomial ops), full GPU suite
set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 > $R/gpurun_out/r2_pytest8.log 2>&1; tail -14 $R/gpurun_out/r2_pytest8.log
