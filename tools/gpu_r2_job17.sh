# round 2, job 17: what a 2^23 / 2^25 MSM (the shards of 2^26 / 2^28 over 8 GPUs) spends its time on
set -x
R=$PWD
mkdir -p $R/gpurun_out
for lg in 20 22 23 24 25; do timeout 200 python tools/gpu_msm_one.py $lg 0 2>&1 | tail -1; done > $R/gpurun_out/r2_msm_sizes.log; cat $R/gpurun_out/r2_msm_sizes.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 23 0 > $R/gpurun_out/r2_tl.log 2>&1); tail -1 $R/gpurun_out/r2_tl.log
cd $R
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 44 > gpurun_out/r2_msm_timeline_2p23.txt 2>&1
grep -v "big_\|scan_" gpurun_out/r2_msm_timeline_2p23.txt | tail -40
rm -rf gpurun_out/prof_tl
