"""The first bucket-sum level of the medium sizes with its two chains on two waves (k_bucket_level1_pipe) against one lane per work
item (k_bucket_level1_lat, tune_tail 10), alternating.    python tools/gpu_msm_level1_ab.py [curve] LG [LG ...]"""
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
    ref = None
    for rep in range(2):
        for tag, join in (("two waves", 0), ("one lane", 10)):
            ctx.tune_tail(join, 0)
            for _ in range(2): out = ctx.invoke(pts, sc)
            reps = 4 if lg >= 24 else 20
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): ctx.invoke(pts, sc)
            wall = (time.perf_counter() - t0) / reps * 1e3
            aff = sppark_amd.to_affine(out, curve)
            if ref is None: ref = aff
            assert (aff == ref).all(), tag
            d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
            print("%s 2^%d %-10s tail %.3f device %.3f wall %.3f" % (curve, lg, tag, d - a - b, d, wall), flush=True)
    del pts, sc
ctx.tune_tail(0, 0)
