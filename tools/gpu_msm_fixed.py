"""Fixed-base mode of the preloaded bases against the plain preloaded path at 2^LG points.
    python tools/gpu_msm_fixed.py [curve] LG[:wbits[,wbits...]] [LG ...]
Prints the time of set_points (plain / with tables), the MSM over the preloaded bases both ways (HIP-event
breakdown and wall clock), and asserts that both give the same point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
args = sys.argv[1:]
only_fixed = "--only-fixed" in args               # (for a kernel trace of the fixed-base MSM alone)
if only_fixed: args.remove("--only-fixed")
curve = args.pop(0) if not args[0][0].isdigit() else "bls12_381"
ctx = sppark_amd.MsmContext(curve); ctx.enable_timing(True)
if os.environ.get("FB_BIG"): ctx.tune_split(int(os.environ["FB_BIG"]))      # partitions above this go to the cooperative sort
for a in args:
    lg, _, wl = a.partition(":")
    lg = int(lg); n = 1 << lg
    widths = [int(w) for w in wl.split(",")] if wl else [0]
    pts, _ = synth.replicated_points(n, curve, 2048, 1)
    sc = synth.uniform_scalars(n, curve, 1)
    def timed(tag):
        out = None
        for _ in range(2):
            out = ctx.invoke(None, sc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5 if lg >= 24 else 20
        for _ in range(reps):
            ctx.invoke(None, sc)
        wall = (time.perf_counter() - t0) / reps * 1e3
        d, acc, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
        print("2^%d %-28s before-acc %.3f accumulate %.3f tail %.3f device %.3f wall %.3f" % (lg, tag, b, acc, d - acc - b, d, wall), flush=True)
        return sppark_amd.to_affine(out, curve)
    ctx.tune()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.set_points(pts)
    print("2^%d set_points plain: %.1f ms" % (lg, (time.perf_counter() - t0) * 1e3), flush=True)
    ref = None if only_fixed else timed("plain (%d windows)" % ctx.plan(n)["windows"])
    for wb in widths:
        ctx.tune(wbits=wb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.set_points(pts, fixed_base=True)
        ctx.tune()
        W = ctx.fixed_base_windows()
        print("2^%d set_points fixed-base wbits=%d: %d windows, %.1f ms" % (lg, wb, W, (time.perf_counter() - t0) * 1e3), flush=True)
        got = timed("fixed-base W=%d" % W)
        if ref is None: ref = got
        assert (got == ref).all(), (lg, wb)
        # plan knobs of the one-window MSM (the tables stay): FB_LB / FB_SLABS / FB_L / FB_K1 = comma lists
        for env, f in (("FB_LB", lambda v: ctx.tune_sort(v)), ("FB_SLABS", lambda v: ctx.tune(nslabs=v)),
                       ("FB_L", lambda v: ctx.tune(L=v)), ("FB_K1", lambda v: ctx.tune_tail(0, v))):
            for v in [int(x) for x in os.environ.get(env, "").split(",") if x]:
                f(v)
                assert (timed("fixed-base W=%d %s=%d" % (W, env[3:], v)) == ref).all(), (lg, wb, env, v)
            ctx.tune(); ctx.tune_sort(0); ctx.tune_tail(0, 0)
    ctx.set_points(None)
    del pts, sc
