"""Window size / run length / bucket chunk around the automatic plan at 2^LG (shard sizes of the multi-GPU
runs): python tools/gpu_msm_plan_sweep.py LG [LG ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
for lg in (int(a) for a in sys.argv[1:]):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1)
    sc = synth.uniform_scalars(n, "bls12_381", 1)
    def run(tag, **kw):
        ctx.tune(**kw)
        for _ in range(2):
            ctx.invoke(pts, sc)
        pl = ctx.plan(n)
        print("2^%d %-18s windows %2d: before-acc %.2f accumulate %.2f device %.2f" % (lg, tag, pl["windows"], ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2)), flush=True)
    run("auto")
    auto_w = min(22, lg - 4)
    for wb in (auto_w - 1, auto_w + 1, auto_w + 2):
        if 8 <= wb <= 24: run("wbits=%d" % wb, wbits=wb)
    for L in (32, 64, 128):
        run("L=%d" % L, L=L)
    for K in (4, 8, 16):
        run("K=%d" % K, K=K)
    for F in (4, 16):
        run("F=%d" % F, F=F)
    del pts, sc
