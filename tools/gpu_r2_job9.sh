set -x
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_poly_gpu.py -m gpu -x -q > $R/gpurun_out/r2_pytest9.log 2>&1; tail -6 $R/gpurun_out/r2_pytest9.log
