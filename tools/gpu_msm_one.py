"""Run BLS12-381 MSM at 2^LG with the given window sizes (for profiling)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
lg = int(sys.argv[1]); wbs = [x for x in sys.argv[2:]]
n = 1 << lg
base = torch.zeros((2048, 96), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3f
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
if os.environ.get("SPPARK_TOP"): ctx.tune_sums(int(os.environ["SPPARK_TOP"]))
for spec in wbs:
    wb, lb = (int(x) for x in (spec.split(":") + ["0"])[:2])
    ctx.tune(wbits=wb); ctx.tune_sort(lb)
    for _ in range(2):
        ctx.invoke(pts, sc)
    print("wbits", wb, "LB", lb, "sort %.2f accum %.2f device %.2f" % (ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2)), flush=True)
