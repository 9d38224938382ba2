"""BLS12-377 G1 MSM and Fr NTT timings (the curve shares the 28-bit-limb bucket pipeline with BLS12-381)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth, NTTInputOutputOrder as Ord
for lg in (22, 26):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, "bls12_377", 2048, 1)
    sc = synth.uniform_scalars(n, "bls12_377", 1)
    ctx = sppark_amd.MsmContext("bls12_377", stream=torch.cuda.current_stream().cuda_stream); ctx.enable_timing(True)
    ctx.invoke(pts, sc)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); ctx.invoke(pts, sc); best = min(best, time.perf_counter() - t)
    print("bls12_377 G1 MSM 2^%d: %.2f ms (%.3e points/s), accumulate %.2f ms, plan %s" % (lg, best * 1e3, n / best, ctx.kernel_ms(1), ctx.plan(n)), flush=True)
    ctx.close(); del pts, sc
s = torch.cuda.current_stream().cuda_stream
x = torch.randint(0, 2**60, ((1 << 22) * 4,), dtype=torch.int64, device="cuda")
for _ in range(3): sppark_amd.NTT(0, x, Ord.NR, "bls12_377", stream=s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): sppark_amd.NTT(0, x, Ord.NR, "bls12_377", stream=s)
e1.record(); torch.cuda.synchronize()
print("bls12_377 Fr NTT 2^22: %.3f ms" % (e0.elapsed_time(e1) / 10))
