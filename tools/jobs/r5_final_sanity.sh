#!/bin/bash
# The last GPU job of round 5: the full GPU suite, smoke() and the default bench line on the committed tree.
R=$PWD; mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $R/gpurun_out/r5f_pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $R/gpurun_out/r5f_pytest_gpu.log | head -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $R/gpurun_out/r5f_bench.json 2> $R/gpurun_out/r5f_bench.err; python -c "
import json; d=json.load(open('$R/gpurun_out/r5f_bench.json')); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['timed_msm_equals_oracle'], d['ntt']['forward_ms'])"
