R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_x
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o msm -- python tools/gpu_msm_one.py 26 0 > $R/gpurun_out/prof_x.log 2>&1)
tail -1 $R/gpurun_out/prof_x.log
