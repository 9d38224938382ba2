"""Run length A/B at arbitrary sizes: the fitted automatic plan against forced run lengths, with the phase times.
    python tools/gpu_msm_L_ab.py N [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
for n in (int(a) for a in sys.argv[1:]):
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1); sc = synth.uniform_scalars(n, "bls12_381", 1)
    ctx.tune(); auto = ctx.plan(n)["run_length"]
    p2 = 1
    while p2 * 2 <= auto: p2 *= 2
    for rep in range(2):
        for L in (0, p2, 2 * p2, auto):
            ctx.tune(L=L)
            for _ in range(3): ctx.invoke(pts, sc)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): ctx.invoke(pts, sc)
            wall = (time.perf_counter() - t) / 10 * 1e3
            d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
            pl = ctx.plan(n)
            print("n=%d L=%s (%d) windows %d: before-acc %.3f accumulate %.3f tail %.3f device %.3f wall %.3f" % (
                n, "auto" if L == 0 else "forced", pl["run_length"], pl["windows"], b, a, d - a - b, d, wall), flush=True)
    del pts, sc
