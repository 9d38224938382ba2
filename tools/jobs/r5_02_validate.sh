#!/bin/bash
# Round 5, job 2: the whole GPU suite on the tree with the adopted G2 wave-pair accumulation, the one-launch small transforms
# and the reference field pin; then the small-transform timing against the reference's build (default library, and a
# tuning build with SPPARK_NTT_SMALL_MAX=0 = the old multi-launch path, same box) and the G2 figures of the default path.
mkdir -p gpurun_out; out=gpurun_out/r5_02
timeout 900 python -m pytest tests -q -x -m gpu --timeout 300 2>&1 | tail -15 | tee $out.pytest.log
timeout 300 python tools/gpu_ntt_small_vs_reference.py 2>&1 | grep -v amdgpu | tee $out.ntt_small.log
echo "== the general path at the same sizes (tuning build, SPPARK_NTT_SMALL_MAX=0)" | tee -a $out.ntt_small.log
SPPARK_LIBDIR=lib_tuning SPPARK_NTT_SMALL_MAX=0 timeout 200 python tools/gpu_ntt_small_vs_reference.py only=ours 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 " | tee -a $out.ntt_small.log
echo "== NN order (the bit reversal folded in)" | tee -a $out.ntt_small.log
timeout 200 python tools/gpu_ntt_small_vs_reference.py order=0 2>&1 | grep -v amdgpu | grep "2^8 \|2^9 \|2^10 \|rows" | tee -a $out.ntt_small.log
timeout 200 python tools/gpu_g2_bench.py 2>&1 | grep -v amdgpu | tee $out.g2.log
