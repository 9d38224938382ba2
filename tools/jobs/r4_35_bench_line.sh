#!/bin/bash
# round 4, job 35: the bench line on the final tree
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r4_35_bench.json 2> gpurun_out/r4_35_bench.err; tail -c 400 gpurun_out/r4_35_bench.json
