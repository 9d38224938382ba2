"""G2 MSM timing (mult_pippenger_fp2_inf, device-resident inputs, one-shot context per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, sppark_amd
import oracle as O
for name, curve in (("bls12_381", O.BLS12_381_G2), ("bn254", O.BN254_G2)):
    fb = O.FP_BYTES[curve]
    base = np.zeros((1024, 2 * fb + 8), dtype=np.uint8)
    base[:, :2 * fb] = O.g1_gen_points(curve, 1024, 11)
    for lg in (16, 18, 20, 21, 22):
        n = 1 << lg
        pts = torch.from_numpy(base[np.arange(n) % 1024]).cuda()
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x1f
        sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print("%s G2 2^%d: %.1f ms  %.3e points/s" % (name, lg, dt * 1e3, n / dt), flush=True)
