#!/bin/bash
# round 4, job 26: kernel traces of small transforms, ours and the reference's build (kernel durations and gaps)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; out=$R/gpurun_out/r4_26_ntt_small_trace.log; : > $out
cd /tmp && export TMPDIR=/tmp
for side in ours ref; do for lg in 8 12 16 18; do
  rm -rf /tmp/prof_t
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python tools/gpu_ntt_small_vs_reference.py only=$side lg=$lg field=gl64 > /tmp/run.log 2>&1)
  echo "== $side gl64 2^$lg: $(grep 'fwd NR' /tmp/run.log)" >> $out
  db=$(find /tmp/prof_t -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" 2>&1 | head -9 >> $out
  python $R/tools/rocprof_timeline.py "$db" 8 2>&1 >> $out
done; done
cat $out | cut -c1-170
