"""Latency of a chain of dependent point operations at the occupancy of the MSM's tail (one work-group per CU):
one wave per operation (add_pairs / dbl_pairs / the single-chain add) against four waves per operation (ec/xyzz_coop.hpp)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import oracle as O
from sppark_amd import ffi
O.build()
L = ffi.load_devtest("bls12_381")
L.sppark_devtest_chain_bench.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
L.sppark_devtest_chain_bench.restype = ffi._Error
fb, n = 48, 64
P = lambda a: a.ctypes.data
A = O.g1_gen_points(0, n, 11)
one = O.field_op(O.FIELD_BLS_FP, 4, O.int_to_limbs(1, fb)).view(np.uint8)
xa = np.zeros((n, 4 * fb), dtype=np.uint8); xa[:, :2 * fb] = A; xa[:, 2 * fb:3 * fb] = one; xa[:, 3 * fb:] = one
tmp = np.zeros_like(xa); xb = np.zeros_like(xa)
ffi.check(L, L.sppark_devtest_xyzz_op(1, P(tmp), P(xa), P(O.g1_gen_points(0, n, 13)), n)); xa = tmp.copy()
ffi.check(L, L.sppark_devtest_xyzz_op(1, P(xb), P(xa), P(O.g1_gen_points(0, n, 14)), n))
out = np.zeros_like(xa)
names = {0: "add_pairs, one wave", 1: "coop_add, four waves", 2: "dbl_pairs, one wave", 3: "coop_dbl, four waves", 4: "add (single chains), one wave"}
res = {}
for nblocks in (1, 256, 512, 1024):
    for mode in (0, 1, 4, 2, 3):
        t = []
        for reps in (16, 48):
            ms = ctypes.c_float()
            ffi.check(L, L.sppark_devtest_chain_bench(mode, reps, nblocks, P(out), P(xa), P(xb), ctypes.byref(ms)))
            t.append(ms.value)
        per = (t[1] - t[0]) / 32 * 1e3
        res[(nblocks, mode)] = per
        print("%4d work-groups  %-32s %7.2f us per operation" % (nblocks, names[mode], per), flush=True)
# the two forms must agree: 48 additions of xb onto xa
o0 = np.zeros_like(xa); o1 = np.zeros_like(xa); ms = ctypes.c_float()
ffi.check(L, L.sppark_devtest_chain_bench(0, 48, 1, P(o0), P(xa), P(xb), ctypes.byref(ms)))
ffi.check(L, L.sppark_devtest_chain_bench(1, 48, 1, P(o1), P(xa), P(xb), ctypes.byref(ms)))
print("coop == one-wave result after 48 additions:", bool((o0 == o1).all()))
