# round 4, job 17: the refined fit rule (only grids above one round, at most two waves per SIMD counted, >= 3 % saving) at every size, both curves; timelines
set -x
R=$PWD; mkdir -p $R/gpurun_out
timeout 600 python tools/gpu_msm_tail.py ab 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 > $R/gpurun_out/r4_17_msm_sizes.log 2>&1; grep -v amdgpu $R/gpurun_out/r4_17_msm_sizes.log | grep "auto "
timeout 300 python tools/gpu_msm_tail.py bn254 ab 14 16 17 18 20 22 23 26 2>&1 | grep "auto " | tee $R/gpurun_out/r4_17_msm_bn254.log
timeout 300 python tools/gpu_msm_tail.py ab 17 18 2>&1 | grep -v amdgpu | tail -8
# non-power-of-two sizes: the fitted plan against the power-of-two run length (forced)
python - <<'PY' 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r4_17_odd_sizes.log
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, sppark_amd
from sppark_amd import synth
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
for n in (100000, 300000, 700000, 1500000, 3000000, 6000000, 12000000, 50000000):
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1); sc = synth.uniform_scalars(n, "bls12_381", 1)
    res = []
    for forced in (0, 1):
        ctx.tune()
        L = ctx.plan(n)["run_length"]
        if forced:
            p2 = 1
            while p2 * 2 <= L: p2 *= 2
            cand = [p2, p2 * 2]
            best = None
            for c in cand:
                ctx.tune(L=c)
                for _ in range(2): ctx.invoke(pts, sc)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(5): ctx.invoke(pts, sc)
                dt = (time.perf_counter() - t) / 5 * 1e3
                best = min(best, (dt, c)) if best else (dt, c)
            res.append("best power of two L=%d: %.3f ms" % (best[1], best[0]))
        else:
            for _ in range(2): ctx.invoke(pts, sc)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): ctx.invoke(pts, sc)
            res.append("fitted L=%d: %.3f ms" % (L, (time.perf_counter() - t) / 5 * 1e3))
    print("n = %d: %s | %s" % (n, res[0], res[1]), flush=True)
    del pts, sc
PY
