"""Run length at the sizes around the 64 -> 128 step of the plan (2^20 .. 2^22), alternating so that drift cancels.
    python tools/gpu_msm_L_mid.py [curve] LG [LG ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
args = sys.argv[1:]
curve = args.pop(0) if not args[0].isdigit() else "bls12_381"
ctx = sppark_amd.MsmContext(curve); ctx.enable_timing(True)
for lg in (int(a) for a in args):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, curve, 2048, 1)
    sc = synth.uniform_scalars(n, curve, 1)
    res = {}
    for rnd in range(4):
        for L in (0, 64, 128, 256):
            ctx.tune(L=L)
            ctx.invoke(pts, sc)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ctx.invoke(pts, sc)
            wall = (time.perf_counter() - t0) / 10 * 1e3
            d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
            res.setdefault(L, []).append((wall, d, a, d - a - b))
    for L, v in res.items():
        w = min(x[0] for x in v); k = min(v)[1:]
        print("%s 2^%d L=%-4s wall(min of 4x10) %.3f | device %.3f accumulate %.3f tail %.3f   (plan L %d)" % (
            curve, lg, L or "auto", w, k[0], k[1], k[2], ctx.plan(n)["run_length"] if L == 0 else L), flush=True)
    ctx.tune()
ctx.close()
