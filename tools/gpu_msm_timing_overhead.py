import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import synth
for lg in (10, 12, 16, 18, 20):
    n = 1 << lg
    pts, _ = synth.replicated_points(n, "bls12_381", 2048, 1)
    sc = synth.uniform_scalars(n, "bls12_381", 1)
    ctx = sppark_amd.MsmContext("bls12_381", device_id=-1, stream=torch.cuda.current_stream().cuda_stream)
    for timing in (True, False, True, False):
        ctx.enable_timing(timing)
        ctx.invoke(pts, sc)
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 50 if lg < 22 else 5
        for _ in range(reps): ctx.invoke(pts, sc)
        torch.cuda.synchronize()
        print("2^%d timing=%s: %.4f ms" % (lg, timing, (time.perf_counter() - t) / reps * 1e3), flush=True)
    ctx.close()
