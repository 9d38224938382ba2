# round 2, job 10: dispatch timeline of one 2^26 MSM (where the 42 ms outside k_accumulate go)
set -x
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/r2_tl.log 2>&1); tail -3 $R/gpurun_out/r2_tl.log
cd $R
python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 45 > gpurun_out/r2_msm_timeline.txt 2>&1
cat gpurun_out/r2_msm_timeline.txt
rm -rf gpurun_out/prof_tl
