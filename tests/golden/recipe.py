"""Deterministic input recipes shared by make_golden.py and the tests.

MSM inputs mirror poc/msm-cuda/src/util.rs:11-38 (2^k distinct points replicated
cyclically, index 3 forced to infinity, uniform scalars) but seeded."""
import numpy as np

import oracle as O


def msm_inputs(curve, n, seed, ndistinct=64, flagged=False, edge=True):
    fb = O.FP_BYTES[curve]
    stride = 2 * fb + 8 if flagged else 2 * fb
    base = O.g1_gen_points(curve, min(ndistinct, max(n, 1)), seed)
    pts = np.zeros((n, stride), dtype=np.uint8)
    if n:
        pts[:, :2 * fb] = base[np.arange(n) % base.shape[0]]
    sc = O.random_scalars(curve, n, seed ^ 0x5ca1a5)
    r = O.FR_MODULUS[curve]
    if edge and n > 3:
        pts[3] = 0                                  # infinity at index 3 (util.rs:24)
        if flagged:
            pts[3, 2 * fb] = 1
            pts[3, :8] = 0xa5                       # garbage coordinates under the flag
    if edge and n > 12:
        sc[5] = 0                                                               # zero scalar
        sc[6] = np.frombuffer((r - 1).to_bytes(32, "little"), dtype=np.uint8)  # r - 1
        sc[7] = sc[8]; pts[7] = pts[8]                                          # same point & scalar (doubling)
        sc[9] = np.frombuffer(((r + 1) // 2).to_bytes(32, "little"), dtype=np.uint8)
        sc[10] = np.frombuffer(((r - 1) // 2).to_bytes(32, "little"), dtype=np.uint8)
        sc[11] = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
        # P and -P with the same scalar cancel
        p = O.FP_MODULUS[curve]
        pts[12] = pts[1]
        base = (p.bit_length() + 63) // 64 * 8              # bytes of one base-field element (G2: two per coordinate)
        for o in range(fb, 2 * fb, base):
            y = int.from_bytes(pts[1, o:o + base].tobytes(), "little")
            pts[12, o:o + base] = np.frombuffer(((p - y) % p).to_bytes(base, "little"), dtype=np.uint8)
        sc[12] = sc[1]
    return pts, sc


def ntt_input(field, lg, seed):
    rng = np.random.default_rng(seed)
    n = 1 << lg
    if field in O.CURVE_ID:                                 # 256-bit scalar field: (n, 4) u64 limbs < r
        r = O.FR_MODULUS[O.CURVE_ID[field]]
        vals = [int.from_bytes(rng.bytes(40), "little") % r for _ in range(n)]
        return np.array([[(v >> (64 * k)) & 0xffffffffffffffff for k in range(4)] for v in vals], dtype=np.uint64)
    if field == "gl64":
        return (rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64)) % np.uint64(O.GL64_P)
    return (rng.integers(0, 1 << 32, size=n, dtype=np.uint64) % O.BB31_P).astype(np.uint32)
