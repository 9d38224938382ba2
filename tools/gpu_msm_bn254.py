"""alt_bn128 G1 MSM phase timings at 2^LG (device-resident inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
base = torch.zeros((2048, 64), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 64, "bn254")
pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x1f
ctx = sppark_amd.MsmContext("bn254"); ctx.enable_timing(True)
for _ in range(3):
    ctx.invoke(pts, sc)
    print("bn254 2^%d sort %.2f accum %.2f device %.2f" % (lg, ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2)), flush=True)
