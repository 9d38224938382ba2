"""The narrow end of the piece tree (small MSMs): one launch for every level of at most 2^x work items (k_piece_tail_coop)
against a launch per level.    python tools/gpu_msm_piece_tail.py [curve] LG [LG ...]"""
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
    ref = [None]
    def run(tag, join):
        ctx.tune_tail(join, 0)
        for _ in range(3):
            out = ctx.invoke(pts, sc)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                ctx.invoke(pts, sc)
            best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
        aff = sppark_amd.to_affine(out, curve)
        if ref[0] is None: ref[0] = aff
        assert (aff == ref[0]).all(), tag
        d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
        print("2^%d %-28s tail %.3f device %.3f wall %.3f" % (lg, tag, d - a - b, d, best), flush=True)
    run("auto", 0)
    run("a launch per level", 8)
    for x in (12, 13, 14, 15, 16, 17, 18, 20):
        run("fused from 2^%d work items" % x, 16 + x)
    run("auto", 0)
    ctx.tune_tail(0, 0)
