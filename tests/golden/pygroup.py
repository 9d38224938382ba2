"""Independent pure-Python big-int group law for G1 and G2 of BLS12-381 / alt_bn128.

Second pin of the MSM oracle (the first is oracle/_ref, the reference's own
msm/pippenger.hpp compiled over the oracle's field class): nothing here touches
oracle/ff.hpp or oracle/ec.hpp -- points are decoded from their wire bytes
(Montgomery, R = 2^(8*sizeof(fp))), checked against the curve equation, and
sum_i s_i * P_i is evaluated with textbook affine chord-and-tangent arithmetic and
double-and-add.  Used by make_golden.py (every golden case with n <= 1024 must agree
with the reference build) and by tests/test_oracle.py.
"""

FP = {
    "bls12_381": int("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab", 16),
    "bn254": int("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", 16),
    "bls12_377": int("01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001", 16),
    "pallas": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
    "vesta": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001,
}
FP_BYTES = {"bls12_381": 48, "bn254": 32, "bls12_377": 48, "pallas": 32, "vesta": 32}
B_G1 = {"bls12_381": 4, "bn254": 3, "bls12_377": 1, "pallas": 5, "vesta": 5}     # y^2 = x^3 + b
FP2_NR = {"bls12_381": 1, "bn254": 1, "bls12_377": 5}   # Fp2 = Fp[u]/(u^2 + NR)
# twists: y^2 = x^3 + b', b' = 4(1+u) (BLS12-381, M-type), 3/(9+u) (alt_bn128, D-type), 1/u (BLS12-377)


def _b_g2(curve):
    p = FP[curve]
    if curve == "bls12_381":
        return (4, 4)
    if curve == "bls12_377":
        return f2_inv((0, 1), p, 5)
    return f2_mul((3, 0), f2_inv((9, 1), p), p)


# ---- Fp2 ---------------------------------------------------------------------
def f2_add(a, b, p): return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)
def f2_sub(a, b, p): return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)
def f2_mul(a, b, p, nr=1): return ((a[0] * b[0] - nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_inv(a, p, nr=1):
    n = pow((a[0] * a[0] + nr * a[1] * a[1]) % p, -1, p)
    return (a[0] * n % p, -a[1] * n % p)


class Fp:
    """field ops on plain ints"""
    def __init__(self, p): self.p = p; self.zero = 0
    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return a * b % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def small(self, k): return k % self.p


class Fp2:
    """field ops on (c0, c1) tuples; u^2 = -nr"""
    def __init__(self, p, nr=1): self.p = p; self.nr = nr; self.zero = (0, 0)
    def add(self, a, b): return f2_add(a, b, self.p)
    def sub(self, a, b): return f2_sub(a, b, self.p)
    def mul(self, a, b): return f2_mul(a, b, self.p, self.nr)
    def inv(self, a): return f2_inv(a, self.p, self.nr)
    def small(self, k): return (k % self.p, 0)


# ---- affine group law (None = infinity) ----------------------------------------
def ec_add(F, P, Q):
    if P is None: return Q
    if Q is None: return P
    if P[0] == Q[0]:
        if P[1] != Q[1] or P[1] == F.zero:
            return None
        lam = F.mul(F.mul(F.small(3), F.mul(P[0], P[0])), F.inv(F.add(P[1], P[1])))
    else:
        lam = F.mul(F.sub(Q[1], P[1]), F.inv(F.sub(Q[0], P[0])))
    x3 = F.sub(F.sub(F.mul(lam, lam), P[0]), Q[0])
    return (x3, F.sub(F.mul(lam, F.sub(P[0], x3)), P[1]))


def ec_mul(F, P, k):
    R = None
    while k:
        if k & 1: R = ec_add(F, R, P)
        P = ec_add(F, P, P); k >>= 1
    return R


def on_curve(F, P, b):
    if P is None: return True
    return F.mul(P[1], P[1]) == F.add(F.mul(F.mul(P[0], P[0]), P[0]), b)


# ---- wire format -----------------------------------------------------------------
def decode_points(curve, g2, raw, stride, flagged):
    """raw: bytes of n records; returns a list of affine points (None = infinity).
    Affine_t: all-zero coordinates = infinity (ec/affine_t.hpp:17-122);
    Affine_inf_t: flag byte after the coordinates."""
    p, fb = FP[curve], FP_BYTES[curve]
    rinv = pow(1 << (8 * fb), -1, p)
    nco = 4 if g2 else 2
    n = len(raw) // stride
    out = []
    for i in range(n):
        rec = raw[i * stride:(i + 1) * stride]
        co = [int.from_bytes(rec[k * fb:(k + 1) * fb], "little") for k in range(nco)]
        if flagged:
            inf = rec[nco * fb] != 0
        else:
            inf = all(c == 0 for c in co)
        if inf:
            out.append(None); continue
        assert all(c < p for c in co), "non-canonical coordinate"
        co = [c * rinv % p for c in co]
        out.append(((co[0], co[1]), (co[2], co[3])) if g2 else (co[0], co[1]))
    return out


def encode_affine(curve, g2, P):
    """affine point -> X | Y wire bytes (Montgomery), infinity -> all-zero"""
    p, fb = FP[curve], FP_BYTES[curve]
    R = 1 << (8 * fb)
    if P is None:
        vals = [0] * (4 if g2 else 2)
    else:
        vals = [P[0][0], P[0][1], P[1][0], P[1][1]] if g2 else [P[0], P[1]]
    return b"".join((v * R % p).to_bytes(fb, "little") for v in vals)


def msm_affine_bytes(curve, g2, points_raw, stride, flagged, scalars_raw):
    """sum_i s_i*P_i, textbook arithmetic only.  points_raw/scalars_raw: bytes."""
    F = Fp2(FP[curve], FP2_NR[curve]) if g2 else Fp(FP[curve])
    b = _b_g2(curve) if g2 else B_G1[curve]
    pts = decode_points(curve, g2, points_raw, stride, flagged)
    # equal points share one scalar multiplication: sum_i s_i P = (sum_i s_i) P, the scalars added as plain integers
    # (the n = 1000 cases repeat 64 points: 64 double-and-add ladders instead of 1000)
    total = {}
    for i, P in enumerate(pts):
        assert on_curve(F, P, b), "input point %d is not on the curve" % i
        s = int.from_bytes(scalars_raw[32 * i:32 * i + 32], "little")
        if P is None or s == 0:
            continue
        total[P] = total.get(P, 0) + s
    acc = None
    for P, s in total.items():
        acc = ec_add(F, acc, ec_mul(F, P, s))
    return encode_affine(curve, g2, acc)
