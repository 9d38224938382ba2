"""Tunable sweep at the headline size (BLS12-381 G1, 2^LG): run length L, slabs, sort split, fan-in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
base = torch.zeros((2048, 96), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3f
ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
def run(tag, **kw):
    lb = kw.pop("LB", 0)
    ctx.tune(**kw); ctx.tune_sort(lb)
    for _ in range(2):
        ctx.invoke(pts, sc)
    print("%-28s sort %.2f accum %.2f device %.2f" % (tag, ctx.kernel_ms(0), ctx.kernel_ms(1), ctx.kernel_ms(2)), flush=True)
run("auto")
for L in (32, 48, 96, 128, 256):
    run("L=%d" % L, L=L)
for ns in (32, 128, 256):
    run("nslabs=%d" % ns, nslabs=ns)
for lb in (7, 8, 10, 11):
    run("LB=%d" % lb, LB=lb)
for F in (4, 16):
    run("F=%d" % F, F=F)
for K in (4, 16, 32):
    run("K=%d" % K, K=K)
run("L=128 K=16", L=128, K=16)
