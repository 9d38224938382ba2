# The adopted cut of the top (plain sum in two pieces from 4096 items): the whole GPU suite, the A/B rows, the bench line.  Outputs: gpurun_out/r6_19_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 600 > $R/gpurun_out/r6_19_pytest_gpu.log 2>&1; grep -n "passed\|failed\|rror" $R/gpurun_out/r6_19_pytest_gpu.log | head -3
timeout 600 python tools/gpu_msm_tail.py ab 16 18 19 20 21 22 26 > $R/gpurun_out/r6_19_top_ab.log 2>&1
grep -v amdgpu $R/gpurun_out/r6_19_top_ab.log | grep "auto\|top per sum"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 900 python bench.py > $R/gpurun_out/r6_19_bench.json 2> $R/gpurun_out/r6_19_bench.err; echo "bench.py wall: $SECONDS s"; tail -c 400 $R/gpurun_out/r6_19_bench.json
