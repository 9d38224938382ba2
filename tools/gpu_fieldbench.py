import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sppark_amd import ffi
for curve in ("bls12_381", "bn254"):
    Lc = ffi.load(curve)
    for op, nm, iters in ((0, "mul", 400), (1, "sqr", 400), (2, "add", 4000), (3, "xyzz madd", 60), (4, "xyzz add", 40)):
        best = 0
        for blocks in (1024, 2048, 4096):
            ms = ctypes.c_float()
            ffi.check(Lc, Lc.sppark_devtest_fieldbench(op, iters, blocks, 256, ctypes.byref(ms)))
            best = max(best, blocks * 256 * iters / (ms.value * 1e-3))
        print("%-10s %-10s %.3e ops/s" % (curve, nm, best), flush=True)
