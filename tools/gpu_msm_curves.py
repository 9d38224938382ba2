"""G1 MSM timings of the curves the bench line does not carry (BLS12-377, Pallas, Vesta), device-resident, uniform scalars: CURVE... [lg...]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
curves = [a for a in sys.argv[1:] if not a.isdigit()] or ["bls12_377", "pallas", "vesta"]
lgs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [22, 26]
for name in curves:
    for lg in lgs:
        n = 1 << lg
        pts, _ = synth.replicated_points(n, name, 2048, 1)
        sc = synth.uniform_scalars(n, name, 1)
        ctx = sppark_amd.MsmContext(name, stream=torch.cuda.current_stream().cuda_stream); ctx.enable_timing(True)
        ctx.invoke(pts, sc)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); ctx.invoke(pts, sc); best = min(best, time.perf_counter() - t)
        print("%s G1 MSM 2^%d: %.2f ms (%.3e points/s), accumulate %.2f ms, windows %d" % (name, lg, best * 1e3, n / best, ctx.kernel_ms(1), ctx.plan(n)["windows"]), flush=True)
        ctx.close(); del pts, sc
