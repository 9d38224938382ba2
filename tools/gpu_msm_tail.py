"""The tail of an MSM at 2^LG points (record list + bucket sums): k_join_runs on / off, run length, fan-in,
the first bucket-sum level's chunk, window size.
    python tools/gpu_msm_tail.py [curve] MODE LG [LG ...]      MODE: ab | sweep | grid | sort
ab: automatic plan and join off only; sweep: one knob at a time around the automatic plan;
grid: (window bits x run length x fan-in) for the small sizes; sort: split of the bucket index between the two
sort levels (low bits) x point slabs -- watch "before-acc"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
args = sys.argv[1:]
curve = args.pop(0) if args[0] not in ("ab", "sweep", "grid", "sort") else "bls12_381"
mode = args.pop(0)
ctx = sppark_amd.MsmContext(curve); ctx.enable_timing(True)
for lg in (int(a) for a in args):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, curve, 2048, 1)
    sc = synth.uniform_scalars(n, curve, 1)
    ref = [None]
    def run(tag, join=0, k1=0, lb=0, **kw):
        ctx.tune(**kw); ctx.tune_tail(join, k1); ctx.tune_sort(lb)
        out = None
        for _ in range(3):
            out = ctx.invoke(pts, sc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5 if lg >= 24 else 20
        for _ in range(reps):
            ctx.invoke(pts, sc)
        wall = (time.perf_counter() - t0) / reps * 1e3
        aff = sppark_amd.to_affine(out, curve)
        if ref[0] is None: ref[0] = aff
        assert (aff == ref[0]).all(), tag
        pl = ctx.plan(n)
        d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
        print("2^%d %-26s windows %2d L %3d: before-acc %.3f accumulate %.3f tail %.3f device %.3f wall %.3f" % (
            lg, tag, pl["windows"], pl["run_length"], b, a, d - a - b, d, wall), flush=True)
        return d
    run("auto")
    if mode != "sort":
        run("join off", join=1)
        run("no low-latency sums", join=3)
        run("no cooperative kernels", join=4)
        run("no piece tree", join=5)
        run("per-lane conversion", join=6)
        run("top per sum", join=7)
    if mode == "sort":
        pl = ctx.plan(n)
        for lb in range(max(1, pl["low_bits"] - 1), min(13, pl["window_bits"] - 1) + 1):
            if pl["window_bits"] - 1 - lb > 12: continue
            run("low bits %d (2^%d partitions)" % (lb, pl["window_bits"] - 1 - lb), lb=lb)
        ctx.tune_sort(0)
        for ns in (4, 8, 16, 32, 64, 128):
            run("slabs %d" % ns, nslabs=ns)
    if mode == "sweep":
        for L in (32, 64, 128, 256):
            run("L=%d" % L, L=L)
        for F in (4, 16):
            run("F=%d" % F, F=F)
        for k1 in (4, 16):
            run("K1=%d" % k1, k1=k1)
        pl = ctx.plan(n)
        ctx.tune(); ctx.tune_tail()
        for wb in sorted({pl["window_bits"] - 1, pl["window_bits"] + 1}):
            if 4 <= wb <= 24: run("wbits=%d" % wb, wbits=wb)
    if mode == "grid":
        ctx.tune(); ctx.tune_tail()
        auto = ctx.plan(n)["window_bits"]
        best = (1e9, "")
        for wb in range(max(4, auto - 1), min(lg, auto + 8)):
            for L in (8, 16, 32, 64):
                for F in (4,):
                    if (n * ((255 + wb - 1) // wb)) // L < 4096:
                        continue
                    d = run("wbits=%d L=%d F=%d" % (wb, L, F), wbits=wb, L=L, F=F)
                    best = min(best, (d, "wbits=%d L=%d F=%d" % (wb, L, F)))
        print("2^%d best: %s %.3f ms" % (lg, best[1], best[0]), flush=True)
    del pts, sc
