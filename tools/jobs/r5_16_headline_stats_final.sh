#!/bin/bash
# Round 5, job 16: rocprofv3 --kernel-trace --stats of the HEADLINE alone on the final tree (bench.py without extras / NTT / CPU
# baseline: every k_accumulate call in the summary is a 2^26-point launch, so its average is the figure roofline.kernel_ms of the
# same run must agree with), then smoke().
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5h
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5h -o bench -- python bench.py --no-extras --no-ntt --no-cpu-baseline > $R/gpurun_out/r5_16_bench_headline.json 2> $R/gpurun_out/r5_16_bench_headline.err)
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r5h -name "*.db" | head -1) > gpurun_out/r5_16_bench_headline_rocprofv3_summary.txt 2>&1
head -24 gpurun_out/r5_16_bench_headline_rocprofv3_summary.txt | cut -c1-130
rm -rf gpurun_out/prof_r5h
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
