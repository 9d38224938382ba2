#!/bin/bash
# Round 5, job 9: (a) rocprofv3 --kernel-trace --stats of the HEADLINE alone (bench.py without extras / NTT / CPU baseline: every
# k_accumulate call in the summary is a 2^26-point launch, so its average is the figure roofline.kernel_ms must agree with);
# (b) the small-transform table against the reference's build on the final tree, all four orders.
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5h
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5h -o bench -- python bench.py --no-extras --no-ntt --no-cpu-baseline > $R/gpurun_out/r5_09_bench_headline.json 2> $R/gpurun_out/r5_09_bench_headline.err)
cd $R
python tools/rocprof_summary.py $(find gpurun_out/prof_r5h -name "*.db" | head -1) > gpurun_out/r5_09_bench_headline_rocprofv3_summary.txt 2>&1
head -24 gpurun_out/r5_09_bench_headline_rocprofv3_summary.txt | cut -c1-130
rm -rf gpurun_out/prof_r5h
out=gpurun_out/r5_09; : > $out.ntt_small.log
for o in 1 0 2 3; do
  echo "== order $o (0 NN, 1 NR, 2 RN, 3 RR)" | tee -a $out.ntt_small.log
  timeout 300 python tools/gpu_ntt_small_vs_reference.py order=$o 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|2^11 \|2^12 \|2^16 \|2^20 \|rows" | tee -a $out.ntt_small.log
done
