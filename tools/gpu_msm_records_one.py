"""Three BLS12-381 MSMs of 2^26 points with 4-byte (packed) or 8-byte (wide) level-A sort records, for a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --stats ... -- python tools/gpu_msm_records_one.py packed|wide [k_lo bits, 0 = automatic]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
n = 1 << 26
base = torch.zeros((2048, 96), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
g = torch.Generator(device="cuda"); g.manual_seed(26)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3f
ctx = sppark_amd.MsmContext("bls12_381")
ctx.tune(nslabs=64 if sys.argv[1] == "wide" else 0)        # (an explicit slab count keeps the wide records; 64 is the automatic one)
if len(sys.argv) > 2:
    ctx.tune_sort(int(sys.argv[2]))                         # the split of the bucket index between the two sort levels
for _ in range(3):
    ctx.invoke(pts, sc)
