# Pieces per subset sum / plain sum of the cooperative top, swept (tuning build of bls12_381).  Outputs: gpurun_out/r6_18_top_cut_sweep.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
SPPARK_LIBDIR=lib_tuning timeout 600 python tools/gpu_msm_top_cut.py 18 19 20 21 22 24 2>&1 | grep -v amdgpu > $R/gpurun_out/r6_18_top_cut_sweep.log
cat $R/gpurun_out/r6_18_top_cut_sweep.log
