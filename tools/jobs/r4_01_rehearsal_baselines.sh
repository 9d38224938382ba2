# round 4, job 1: configs[3] rehearsal at full size on one GPU, the new tests, "before" evidence for the
# BabyBear NTT (PMC) and the small-MSM tail (timelines), and one short bench.py to see the new keys of the line
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 900 python tools/gpu_config3_rehearsal.py a b c > $R/gpurun_out/r4_01_config3_rehearsal.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_01_config3_rehearsal.log | cut -c1-400 | tail -20
timeout 900 python -m pytest tests/test_bench_multirank_gpu.py tests/test_msm_gpu.py -m gpu -x -q -k "multirank or above_2p28" > $R/gpurun_out/r4_01_pytest.log 2>&1; tail -5 $R/gpurun_out/r4_01_pytest.log
cd /tmp && export TMPDIR=/tmp; cd $R
rm -f $R/gpurun_out/pmc_ntt_bb31.txt $R/gpurun_out/pmc_ntt_bb31_nn.txt
bash tools/gpu_pmc_job.sh ntt_bb31 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD|FETCH_SIZE|WRITE_SIZE" -- python tools/gpu_ntt_one.py bb31 24 6
bash tools/gpu_pmc_job.sh ntt_bb31_nn "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY|FETCH_SIZE|WRITE_SIZE" -- python tools/gpu_ntt_one.py bb31 24 6 NN | grep -i "bitrev\|kernel "
bash tools/gpu_pmc_job.sh ntt_gl64_nn "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" -- python tools/gpu_ntt_one.py gl64 24 6 NN | grep -i "bitrev\|kernel "
for lg in 16 18; do
  rm -rf gpurun_out/prof_tl
  (cd /tmp && cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o tl -- python tools/gpu_msm_one.py $lg 0 > $R/gpurun_out/r4_01_tl.log 2>&1)
  python tools/rocprof_timeline.py $(find gpurun_out/prof_tl -name "*.db" | head -1) 48 > gpurun_out/r4_01_msm_timeline_2p${lg}_before.txt 2>&1
  tail -40 gpurun_out/r4_01_msm_timeline_2p${lg}_before.txt | cut -c1-130
done
rm -rf gpurun_out/prof_tl
timeout 900 python bench.py --steps 5 --warmup 1 > $R/gpurun_out/r4_01_bench.json 2> $R/gpurun_out/r4_01_bench.err; tail -c 300 $R/gpurun_out/r4_01_bench.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4_01_bench.json") if l.startswith("{")][0])
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_min")})
print("ntt", {k: d["ntt"][k] for k in ("forward_ms", "inverse_ms", "forward_nn_ms")}, d["ntt"]["through_ffi"])
e = d["extras"]
for k in ("h2d_peak_gbs", "d2h_peak_gbs", "h2d_pageable_gbs", "babybear_ntt_through_ffi", "alt_bn128_g1_msm_roofline", "mult_pippenger_inf_host_buffers", "shard_sizes"):
    print(k, e.get(k))
print("cpu", d["cpu_baseline"])
PY
du -sh gpurun_out
