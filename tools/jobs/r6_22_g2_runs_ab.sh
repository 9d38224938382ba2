# G2 run lengths: the tuning library built BEFORE the rule (its "auto" = the old plan, its L columns forced) against the product library with the rule
# (every column = the new automatic plan), same box.  Outputs: gpurun_out/r6_22_g2_runs_ab.log
set -x
R=$PWD; mkdir -p $R/gpurun_out
(echo "== library before the rule (tuning build; auto = old plan)"; SPPARK_LIBDIR=lib_tuning timeout 600 python tools/gpu_g2_L.py 16 18 20 21 2>&1 | grep -v amdgpu | cut -c1-100
 echo "== product library with the rule (no knobs: every column is the automatic plan)"; timeout 600 python tools/gpu_g2_L.py 16 18 20 21 2>&1 | grep -v amdgpu | cut -c1-100) > $R/gpurun_out/r6_22_g2_runs_ab.log
cat $R/gpurun_out/r6_22_g2_runs_ab.log
