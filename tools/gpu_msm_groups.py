"""Window-group sweep of the two-stream MSM pipeline at 2^LG (BLS12-381 and alt_bn128):
per group count: exposed time before the first accumulation, summed k_accumulate time, device total,
wall time of one invoke.  python tools/gpu_msm_groups.py LG [groups ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
lg = int(sys.argv[1]); groups = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4, 6, 12]
n = 1 << lg
for curve in ("bls12_381", "bn254"):
    pts, _ = synth.replicated_points(n, curve, 2048, 1)
    sc = synth.uniform_scalars(n, curve, 1)
    ctx = sppark_amd.MsmContext(curve, stream=torch.cuda.current_stream().cuda_stream); ctx.enable_timing(True)
    for g in groups:
        ctx.tune_pipeline(groups=g)
        ctx.invoke(pts, sc)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); ctx.invoke(pts, sc); best = min(best, time.perf_counter() - t)
        print("%s 2^%d groups %2d (used %d): exposed %.2f  accumulate %.2f  device %.2f  wall %.2f ms  scratch %.1f GB"
              % (curve, lg, g, int(ctx.kernel_ms(3)), ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2), best * 1e3, ctx.scratch_bytes() / 1e9), flush=True)
    ctx.close(); del pts, sc
