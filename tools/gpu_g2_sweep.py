"""One-knob-at-a-time sweep of the G2 MSM's plan (a TUNING build: SPPARK_LIBDIR=lib_tuning; the knobs are the SPPARK_G2_*
variables of mult_pippenger_fp2_inf, read on every call).  The plan rules were tuned on G1, whose additions cost a third.
    SPPARK_LIBDIR=lib_tuning python tools/gpu_g2_sweep.py [curve] LG [LG ...]"""
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
KNOBS = ("WBITS", "L", "F", "K", "K1", "TOP", "JOIN")
for lg in (int(a) for a in args):
    n = 1 << lg
    pts = torch.from_numpy(base[np.arange(n) % 1024]).cuda()
    sc = synth.uniform_scalars(n, name, 1)
    ref = [None]
    def run(tag, **kw):
        for k in KNOBS:
            os.environ.pop("SPPARK_G2_" + k, None)
        for k, v in kw.items():
            os.environ["SPPARK_G2_" + k] = str(v)
        out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
        torch.cuda.synchronize(); t = time.perf_counter()
        reps = 3 if lg >= 22 else 6
        for _ in range(reps):
            out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        aff = sppark_amd.to_affine_g2(out, name)
        if ref[0] is None: ref[0] = aff
        assert (aff == ref[0]).all(), tag
        print("%s G2 2^%d %-12s %.2f ms" % (name, lg, tag, dt * 1e3), flush=True)
    run("auto"); run("auto again")
    for top in (1, 256, 512, 1024, 2048): run("TOP=%d" % top, TOP=top)
    for k1 in (2, 4, 8, 16): run("K1=%d" % k1, K1=k1)
    for k in (2, 8): run("K=%d" % k, K=k)
    for L in (32, 64, 128, 256): run("L=%d" % L, L=L)
    for f in (4, 8, 32): run("F=%d" % f, F=f)
    for wb in (15, 16, 17, 18, 19, 20): run("WBITS=%d" % wb, WBITS=wb)
    run("JOIN=1", JOIN=1)
    run("auto last")
