"""One BLS12-381 (or bn254 / bls12_377) G2 MSM at 2^LG a few times, device-resident (the command that is profiled): CURVE LG REPS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sppark_amd
import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "bls12_381"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 22
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
curve = {"bls12_381": O.BLS12_381_G2, "bn254": O.BN254_G2, "bls12_377": O.BLS12_377_G2}[name]
fb = O.FP_BYTES[curve]
base = np.zeros((1024, 2 * fb + 8), dtype=np.uint8)
base[:, :2 * fb] = O.g1_gen_points(curve, 1024, 11)
n = 1 << lg
pts = torch.from_numpy(base[np.arange(n) % 1024]).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x1f
for _ in range(reps):
    sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
torch.cuda.synchronize()
