"""Run-length (L) / fan-in (F) / bucket-chunk (K) sweep of the MSM at 2^LG: python tools/gpu_msm_L.py LG"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
lg = int(sys.argv[1]); n = 1 << lg
pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1)
sc = synth.uniform_scalars(n, "bls12_381", 1)
ctx = sppark_amd.MsmContext("bls12_381", stream=torch.cuda.current_stream().cuda_stream); ctx.enable_timing(True)
ref = None
for L, F, K in ((0, 0, 0), (64, 8, 8), (96, 8, 8), (128, 8, 8), (192, 8, 8), (256, 8, 8), (128, 4, 8), (128, 16, 8), (64, 8, 4), (64, 8, 16), (128, 8, 16)):
    ctx.tune(L=L, F=F, K=K)
    ctx.invoke(pts, sc)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); out = ctx.invoke(pts, sc); best = min(best, time.perf_counter() - t)
    a = sppark_amd.to_affine(out); ref = a if ref is None else ref
    print("2^%d L %3d F %2d K %2d: before-acc %.2f  accumulate %.2f  device %.2f  wall %.2f ms %s"
          % (lg, L, F, K, ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2), best * 1e3, "OK" if (a == ref).all() else "MISMATCH"), flush=True)
