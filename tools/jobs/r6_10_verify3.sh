#!/bin/bash
# Round 6, job 10: the whole GPU suite (piece tree sized for the top window's dense bucket), the G2 plan sweep with the knobs
# reset between runs.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -m gpu -x -q --timeout 900 > $O/r6_10_pytest.log 2>&1; grep -n "passed\|failed\|rror" $O/r6_10_pytest.log | head -5
SPPARK_LIBDIR=lib_tuning timeout 1500 python $R/tools/gpu_g2_sweep.py 20 22 > $O/r6_10_g2_sweep.log 2>&1
grep "G2" $O/r6_10_g2_sweep.log
