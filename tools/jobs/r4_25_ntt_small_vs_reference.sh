#!/bin/bash
# round 4, job 25: small transforms against the reference's build, and a kernel trace of each side at 2^16 / 2^12
mkdir -p gpurun_out
timeout 300 python tools/gpu_ntt_small_vs_reference.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_25_ntt_small.log
cd /tmp && export TMPDIR=/tmp
for side in ours ref; do for lg in 12 16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${side}_$lg -o t -- python $GRAFT_REPO_ROOT/tools/gpu_ntt_small_vs_reference.py only=$side lg=$lg field=gl64 > /dev/null 2>&1
  echo "== $side gl64 2^$lg: kernel stats" >> $GRAFT_REPO_ROOT/gpurun_out/r4_25_ntt_small.log
  f=$(find /tmp/prof_${side}_$lg -name "*kernel_stats.csv" | head -1)
  head -8 "$f" | cut -d, -f1-7 >> $GRAFT_REPO_ROOT/gpurun_out/r4_25_ntt_small.log
done; done
tail -60 $GRAFT_REPO_ROOT/gpurun_out/r4_25_ntt_small.log
