"""Window / fan-in / chunk sweep for small and mid-size BLS12-381 MSMs (device ms of the 2nd invoke)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
lgs = [int(x) for x in sys.argv[1:]] or list(range(10, 23, 2))
base = torch.zeros((2048, 96), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
for lg in lgs:
    n = 1 << lg
    pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3f
    res = []
    for wb in range(max(4, lg - 10), min(22, max(6, lg - 3)) + 1):
        for F, K in ((32, 8), (8, 8), (8, 4), (16, 4)):
            ctx.tune(wbits=wb, F=F, K=K)
            for _ in range(2):
                ctx.invoke(pts, sc)
            res.append((ctx.kernel_ms(2), wb, F, K))
    ctx.tune()
    for _ in range(2):
        ctx.invoke(pts, sc)
    auto = ctx.kernel_ms(2)
    res.sort()
    print("2^%d auto %.2f ms | best: %s" % (lg, auto, "  ".join("%.2f(c=%d F=%d K=%d)" % r for r in res[:6])), flush=True)
