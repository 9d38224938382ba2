R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_small
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_small -o msm -- python tools/gpu_msm_one.py 16 0 > $R/gpurun_out/prof_small.log 2>&1)
tail -1 $R/gpurun_out/prof_small.log
