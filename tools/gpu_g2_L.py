"""Run length x window bits of the G2 MSM at the medium sizes (tuning build, SPPARK_G2_L / SPPARK_G2_WBITS).
    SPPARK_LIBDIR=lib_tuning python tools/gpu_g2_L.py [curve] LG [LG ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, sppark_amd
import oracle as O
from sppark_amd import synth
args = sys.argv[1:]
name = args.pop(0) if not args[0].isdigit() else "bls12_381"
curve = {"bls12_381": O.BLS12_381_G2, "bn254": O.BN254_G2, "bls12_377": O.BLS12_377_G2}[name]
fb = O.FP_BYTES[curve]
base = np.zeros((1024, 2 * fb + 8), dtype=np.uint8)
base[:, :2 * fb] = O.g1_gen_points(curve, 1024, 11)
for lg in (int(a) for a in args):
    n = 1 << lg
    pts = torch.from_numpy(base[np.arange(n) % 1024]).cuda()
    sc = synth.uniform_scalars(n, name, 1)
    ref = [None]
    def run(tag, **kw):
        for k in ("WBITS", "L"):
            os.environ.pop("SPPARK_G2_" + k, None)
        for k, v in kw.items():
            if v: os.environ["SPPARK_G2_" + k] = str(v)
        out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
        best = 1e9
        for _ in range(5 if lg >= 21 else 8):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        aff = sppark_amd.to_affine_g2(out, name)
        if ref[0] is None: ref[0] = aff
        assert (aff == ref[0]).all(), tag
        return best * 1e3
    row = "%s G2 2^%d  auto %.2f |" % (name, lg, run("auto"))
    for wb in (0, -1, 1):
        for L in (32, 64, 128, 256, 512):
            w = 0
            if wb:
                # automatic width of this size, shifted
                w = {16: 12, 17: 13, 18: 14, 19: 15, 20: 16, 21: 16, 22: 17}.get(lg, 16) + wb
            row += " %sL%d %.2f" % (("w%+d " % wb) if wb else "", L, run("x", L=L, WBITS=w))
        row += " |"
    print(row + " auto %.2f" % run("auto"), flush=True)
