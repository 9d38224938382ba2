"""MSM timing under skewed scalar distributions (SURVEY 8(d) robustness cases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
base = torch.zeros((2048, 96), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3f
cases = {"uniform": sc}
eq = sc.clone(); eq[:] = sc[0]; cases["all equal"] = eq
half = sc.clone(); half[::2] = 0; cases["50% zeros"] = half
s16 = torch.zeros_like(sc); s16[:, :2] = sc[:, :2]; cases["16-bit scalars"] = s16
ones = torch.zeros_like(sc); ones[:, 0] = 1; cases["all ones"] = ones
same = pts.clone(); same[:] = pts[0]
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
for name, s in cases.items():
    for pname, p in (("distinct pts", pts), ("same point", same)):
        if pname == "same point" and name not in ("uniform", "all equal"): continue
        ctx.invoke(p, s); torch.cuda.synchronize(); t = time.perf_counter(); ctx.invoke(p, s); torch.cuda.synchronize()
        print("2^%d %-15s %-12s wall %8.2f ms  sort %7.2f accum %7.2f device %8.2f" % (lg, name, pname, (time.perf_counter() - t) * 1e3, ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2)), flush=True)
