# The subset-sum top with a work-group per PIECE of a sum: parity (the tests that reach the top), then A/B against the per-sum
# form (tail variant 7) at the sizes whose top has 2048 / 4096 items.  Outputs: gpurun_out/r6_17_*
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -x -q -k "bucket_sum_top or tail_variants or msm_vs_oracle or golden_vectors or randomised or large_linearity or piece_tree" > $R/gpurun_out/r6_17_pytest.log 2>&1; tail -3 $R/gpurun_out/r6_17_pytest.log
timeout 600 python tools/gpu_msm_tail.py ab 17 18 19 20 21 22 23 24 26 > $R/gpurun_out/r6_17_top_pieces_ab.log 2>&1
grep -v amdgpu $R/gpurun_out/r6_17_top_pieces_ab.log | grep "auto\|top per sum"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
(cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py 20 0 > $R/gpurun_out/r6_tl.log 2>&1)
(cd $R && python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 24 > gpurun_out/r6_17_tl_2p20.txt 2>&1)
rm -rf $R/gpurun_out/prof_tl
tail -14 $R/gpurun_out/r6_17_tl_2p20.txt | cut -c1-170
