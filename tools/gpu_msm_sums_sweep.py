"""Bucket-sum knobs at the small and medium sizes: buckets per work item of the first level (K1), chunk of the later levels (K),
items per window handed to the subset-sum top.    python tools/gpu_msm_sums_sweep.py LG [LG ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
for lg in (int(a) for a in sys.argv[1:]):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1)
    sc = synth.uniform_scalars(n, "bls12_381", 1)
    ref = [None]
    def run(tag, k1=0, top=0, K=0):
        ctx.tune(K=K); ctx.tune_tail(0, k1); ctx.tune_sums(top)
        for _ in range(3): out = ctx.invoke(pts, sc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ctx.invoke(pts, sc)
        wall = (time.perf_counter() - t0) / 20 * 1e3
        aff = sppark_amd.to_affine(out)
        if ref[0] is None: ref[0] = aff
        assert (aff == ref[0]).all(), tag
        d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
        print("2^%d %-22s tail %.3f device %.3f wall %.3f" % (lg, tag, d - a - b, d, wall), flush=True)
    run("auto")
    for k1 in (1, 2, 4, 8, 16):
        for top in (0, 32, 256, 1):
            for K in (0, 8):
                try: run("K1=%d top=%d K=%d" % (k1, top, K), k1, top, K)
                except Exception as e: print("2^%d K1=%d top=%d K=%d: %s" % (lg, k1, top, K, e))
