"""Generate tests/golden/*.json.  Run in the build container only:

    python tests/golden/make_golden.py

MSM expectations come from the REFERENCE's own msm/pippenger.hpp compiled in
place (oracle/_ref/libref_msm.so, see oracle/ref_shim.cpp), 1-thread and
8-thread paths required to agree -- and, for every case with n <= 1024 (G1 and
G2, both curves, all edge cases), a second time from tests/golden/pygroup.py, a
pure-Python big-int affine group law that shares no code with oracle/ (points are
decoded from their wire bytes and checked against the curve equation there); the
two must agree or generation fails.  Cases so checked carry "python_checked".  NTT expectations come from an independent
definition-level Python big-int DFT written here (the reference has no CPU NTT),
applied through the order/direction/coset semantics of ntt/ntt.cuh:161-213.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle as O          # noqa: E402
import recipe               # noqa: E402
import pygroup              # noqa: E402  (independent pure-Python group law: the second pin)

HERE = os.path.dirname(os.path.abspath(__file__))


def hexs(a):
    return np.ascontiguousarray(a).tobytes().hex()


def python_pin(cname, g2, pts, sc, flagged):
    return pygroup.msm_affine_bytes(cname, g2, pts.tobytes(), pts.shape[1], flagged, sc.tobytes())


def edge_cases(curve, cname, g2):
    """Small named cases with their bytes stored: what the reference's tests and SURVEY 8(c) list
    (zero scalars, r-1, infinity inputs, duplicated / all-equal points, P and -P, everything
    cancelling).  Each expectation = reference build == pure-Python law."""
    r = O.FR_MODULUS[curve]
    le = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
    out = []

    def emit(name, pts, sc, flagged):
        e = O.ref_msm_affine(curve, pts, sc, nthreads=0)
        assert (e == O.ref_msm_affine(curve, pts, sc, nthreads=8)).all(), name
        assert python_pin(cname, g2, pts, sc, flagged) == e.tobytes(), (cname, g2, name)
        out.append({"curve": cname, "n": int(pts.shape[0]), "name": name, "flagged": flagged, "python_checked": True,
                    "points": hexs(pts), "scalars": hexs(sc), "expect_affine": hexs(e)})
    for flagged in (False, True):
        tag = "_flagged" if flagged else ""
        pts, sc = recipe.msm_inputs(curve, 13, 0xed6e + flagged, ndistinct=13, flagged=flagged)       # all edge rows of the recipe
        emit("recipe_edges_13" + tag, pts, sc, flagged)
        pts, sc = recipe.msm_inputs(curve, 24, 0xed6f, ndistinct=24, flagged=flagged, edge=False)
        z = sc.copy(); z[:] = 0
        emit("all_zero_scalars" + tag, pts, z, flagged)                                  # -> infinity
        inf = pts.copy(); inf[:] = 0
        if flagged:
            inf[:, -8] = 1
        emit("all_infinity_points" + tag, inf, sc, flagged)                              # -> infinity
        same = pts.copy(); same[:] = pts[0]
        emit("all_equal_points" + tag, same, sc, flagged)
        eq = sc.copy(); eq[:] = sc[1]
        emit("all_equal_scalars" + tag, pts, eq, flagged)
        both_sc = sc.copy(); both_sc[:] = le(r - 1)
        emit("all_equal_points_scalar_r_minus_1" + tag, same, both_sc, flagged)         # repeated doubling branch
        # pairs (s, P), (r - s, P): the whole sum cancels
        can = pts.copy(); can_sc = sc.copy()
        for i in range(0, 24, 2):
            can[i + 1] = can[i]
            can_sc[i + 1] = le((r - int.from_bytes(can_sc[i].tobytes(), "little")) % r)
        emit("cancelling_pairs" + tag, can, can_sc, flagged)
        one = sc.copy(); one[:] = le(1)
        emit("all_ones_scalars" + tag, pts, one, flagged)                                # plain sum of the points
        top = sc.copy()
        for i in range(24):
            top[i] = le(r - 1 - i)
        emit("scalars_near_r" + tag, pts, top, flagged)
        small = sc.copy(); small[:, 1:] = 0
        emit("8bit_scalars" + tag, pts, small, flagged)
    return out


def make_msm():
    assert O.ref_available(), "oracle/_ref not built (needs /root/reference)"
    cases = []
    for curve, cname in ((O.BLS12_381, "bls12_381"), (O.BN254, "bn254"), (O.BLS12_377, "bls12_377"), (O.PALLAS, "pallas"), (O.VESTA, "vesta")):
        for n, flagged in ((1, False), (2, False), (4, True), (31, False), (32, True), (33, False),
                           (1000, True), (1024, False), (65536 if curve == O.BLS12_381 else 4096, False)):
            seed = 0x5eed5eed0001 + n
            pts, sc = recipe.msm_inputs(curve, n, seed, ndistinct=2048 if n > 4096 else 64, flagged=flagged)
            e1 = O.ref_msm_affine(curve, pts, sc, nthreads=0)
            e8 = O.ref_msm_affine(curve, pts, sc, nthreads=8)
            assert (e1 == e8).all()
            case = {"curve": cname, "n": n, "seed": seed, "flagged": flagged,
                    "ndistinct": 2048 if n > 4096 else 64, "expect_affine": hexs(e1)}
            if n <= 1024:
                assert python_pin(cname, False, pts, sc, flagged) == e1.tobytes(), (cname, n)
                case["python_checked"] = True
            if n <= 33:
                case["points"] = hexs(pts); case["scalars"] = hexs(sc)
            cases.append(case)
        cases += edge_cases(curve, cname, False)
    # KAT of SURVEY Appendix A.4: sum_{i=1..4} i*(i*G) = 30*G
    G = O.g1_generator(O.BLS12_381)
    pts = np.stack([O.g1_mul(O.BLS12_381, G, i) for i in range(1, 5)])
    sc = np.stack([np.frombuffer(int(i).to_bytes(32, "little"), dtype=np.uint8) for i in range(1, 5)])
    e = O.ref_msm_affine(O.BLS12_381, pts, sc, nthreads=0)
    cases.append({"curve": "bls12_381", "n": 4, "kat": "30G", "points": hexs(pts), "scalars": hexs(sc),
                  "flagged": False, "expect_affine": hexs(e)})
    json.dump(cases, open(os.path.join(HERE, "msm_golden.json"), "w"), indent=0)
    print("msm cases:", len(cases))


# ---- G2 (coordinates in Fp2 = Fp[u]/(u^2+1)) ------------------------------------------
# independent Python big-int group law on the twist, used for the KAT sum i*(i*G2) = 30*G2
def fp2_mul(a, b, p): return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)
def fp2_sub(a, b, p): return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)
def fp2_inv(a, p):
    n = pow((a[0] * a[0] + a[1] * a[1]) % p, p - 2, p)
    return (a[0] * n % p, -a[1] * n % p)


def g2_add(P, Q, p):                                       # affine, None = infinity
    if P is None: return Q
    if Q is None: return P
    if P[0] == Q[0]:
        if P[1] != Q[1] or P[1] == (0, 0): return None
        xx = fp2_mul(P[0], P[0], p)
        lam = fp2_mul(((3 * xx[0]) % p, (3 * xx[1]) % p), fp2_inv(((2 * P[1][0]) % p, (2 * P[1][1]) % p), p), p)
    else:
        lam = fp2_mul(fp2_sub(Q[1], P[1], p), fp2_inv(fp2_sub(Q[0], P[0], p), p), p)
    x3 = fp2_sub(fp2_sub(fp2_mul(lam, lam, p), P[0], p), Q[0], p)
    return (x3, fp2_sub(fp2_mul(lam, fp2_sub(P[0], x3, p), p), P[1], p))


def g2_mul(P, k, p):
    R = None
    while k:
        if k & 1: R = g2_add(R, P, p)
        P = g2_add(P, P, p); k >>= 1
    return R


G2_GEN = {
    "bls12_381": ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                   0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
                  (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                   0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
    "bn254": ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
               11559732032986387107991004021392285783925812861821192530917403151452391805634),
              (8495653923123431417604973247489272438418190587263600148770280649306958101930,
               4082367875863433681332203403145435568316851327593401208105741076214120093531)),
}


def g2_wire(P, p, base):
    """affine point -> X.c0|X.c1|Y.c0|Y.c1 Montgomery limbs (R = 2^(8*base))"""
    R = 1 << (8 * base)
    vals = (0, 0, 0, 0) if P is None else (P[0][0], P[0][1], P[1][0], P[1][1])
    return np.frombuffer(b"".join((v * R % p).to_bytes(base, "little") for v in vals), dtype=np.uint8)


def make_msm_g2():
    assert O.ref_available(), "oracle/_ref not built (needs /root/reference)"
    cases = []
    for curve, cname, base in ((O.BLS12_381_G2, "bls12_381", 48), (O.BN254_G2, "bn254", 32), (O.BLS12_377_G2, "bls12_377", 48)):
        p = O.FP_MODULUS[curve]
        if cname in G2_GEN:
            assert (O.g1_generator(curve) == g2_wire(G2_GEN[cname], p, base)).all()
        for n, flagged in ((1, True), (2, False), (31, True), (33, True), (1000, True), (4096, False)):
            seed = 0x5eed5eed0101 + n
            pts, sc = recipe.msm_inputs(curve, n, seed, ndistinct=64, flagged=flagged)
            e1 = O.ref_msm_affine(curve, pts, sc, nthreads=0)
            e8 = O.ref_msm_affine(curve, pts, sc, nthreads=8)
            assert (e1 == e8).all()
            case = {"curve": cname, "n": n, "seed": seed, "flagged": flagged, "ndistinct": 64, "expect_affine": hexs(e1)}
            if n <= 1024:
                assert python_pin(cname, True, pts, sc, flagged) == e1.tobytes(), (cname, n)
                case["python_checked"] = True
            if n <= 33:
                case["points"] = hexs(pts); case["scalars"] = hexs(sc)
            cases.append(case)
        cases += edge_cases(curve, cname, True)
        # KAT by the independent Python group law: sum_{i=1..4} i*(i*G2) = 30*G2
        F2 = pygroup.Fp2(p, pygroup.FP2_NR[cname])
        gen = pygroup.decode_points(cname, True, O.g1_generator(curve).tobytes(), 4 * base, False)[0]
        assert pygroup.on_curve(F2, gen, pygroup._b_g2(cname))
        pts = np.stack([np.frombuffer(pygroup.encode_affine(cname, True, pygroup.ec_mul(F2, gen, i)), dtype=np.uint8) for i in range(1, 5)])
        sc = np.stack([np.frombuffer(int(i).to_bytes(32, "little"), dtype=np.uint8) for i in range(1, 5)])
        e = np.frombuffer(pygroup.encode_affine(cname, True, pygroup.ec_mul(F2, gen, 30)), dtype=np.uint8)
        assert (O.ref_msm_affine(curve, pts, sc, nthreads=0) == e).all()
        cases.append({"curve": cname, "n": 4, "kat": "30*G2 (Python big-int group law)", "points": hexs(pts),
                      "scalars": hexs(sc), "flagged": False, "expect_affine": hexs(e)})
    json.dump(cases, open(os.path.join(HERE, "msm_g2_golden.json"), "w"), indent=0)
    print("msm g2 cases:", len(cases))


# ---- independent big-int NTT (definition level) -----------------------------
def bitrev(i, lg):
    return int(format(i, "0%db" % lg)[::-1], 2) if lg else 0


def py_ntt(a, lg, order, direction, typ, p, top_root, two_adicity, gen):
    n = 1 << lg
    w = pow(top_root, 1 << (two_adicity - lg), p)
    g = gen
    if direction == 1:
        w = pow(w, p - 2, p); g = pow(g, p - 2, p)
    a = list(a)
    B = [bitrev(i, lg) for i in range(n)]
    # SURVEY Appendix A.8 table
    if order == O.RN:
        a = [a[B[i]] for i in range(n)]                      # input is bit-reversed
    if typ == 1 and direction == 0:
        a = [a[j] * pow(g, B[j] if order == O.RR else j, p) % p for j in range(n)]
    X = [sum(a[j] * pow(w, j * k, p) for j in range(n)) % p for k in range(n)]
    if direction == 1:
        ninv = pow(n, p - 2, p)
        X = [x * ninv % p for x in X]
    if typ == 1 and direction == 1:
        X = [X[j] * pow(g, B[j] if order == O.RR else j, p) % p for j in range(n)]
    if order == O.NR:
        X = [X[B[i]] for i in range(n)]                      # output bit-reversed
    return X


def make_ntt():
    cases = []
    R = 1 << 32
    for field in ("gl64", "bb31"):
        for lg in (1, 3, 5):
            x = recipe.ntt_input(field, lg, 0x5eed5eed0002 + lg)
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        if field == "gl64":
                            y = py_ntt([int(v) for v in x], lg, order, direction, typ, O.GL64_P,
                                       0x185629dcda58878c, 32, 7)
                            out = np.array(y, dtype=np.uint64)
                        else:
                            p = O.BB31_P
                            rinv = pow(R, p - 2, p)
                            top = 0x1ffffedc * rinv % p                  # Montgomery -> canonical
                            xc = [int(v) * rinv % p for v in x]
                            y = py_ntt(xc, lg, order, direction, typ, p, top, 27, 3)
                            out = np.array([v * R % p for v in y], dtype=np.uint32)
                        cases.append({"field": field, "lg": lg, "order": order, "direction": direction,
                                      "type": typ, "input": hexs(x), "expect": hexs(out)})
    # 256-bit scalar fields (wire format: Montgomery, R = 2^256); roots g^((r-1)/2^S), g = 7 / 5
    R256 = 1 << 256
    for field, curve, gen, S in (("bls12_381", O.BLS12_381, 7, 32), ("bn254", O.BN254, 5, 28), ("bls12_377", O.BLS12_377, 22, 47),
                                 ("pallas", O.PALLAS, 5, 32), ("vesta", O.VESTA, 5, 32)):
        p = O.FR_MODULUS[curve]
        rinv = pow(R256, p - 2, p)
        top = pow(gen, (p - 1) >> S, p)
        for lg in (1, 3, 4):
            x = recipe.ntt_input(field, lg, 0x5eed5eed0003 + lg)
            xc = [int.from_bytes(row.tobytes(), "little") * rinv % p for row in x]
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        y = py_ntt(xc, lg, order, direction, typ, p, top, S, gen)
                        out = np.frombuffer(b"".join((v * R256 % p).to_bytes(32, "little") for v in y), dtype=np.uint64)
                        cases.append({"field": field, "lg": lg, "order": order, "direction": direction,
                                      "type": typ, "input": hexs(x), "expect": hexs(out)})
    json.dump(cases, open(os.path.join(HERE, "ntt_golden.json"), "w"), indent=0)
    print("ntt cases:", len(cases))


# ---- LDE (NTT::LDE_aux, ntt/ntt.cuh:283-336), definition level ---------------------
# coefficients c = inverse DFT of the evaluations; output[k] = sum_j c_j (g * w_ext^k)^j
def py_lde(x, lg, lgb, p, top_root, two_adicity, gen):
    n = 1 << lg
    w = pow(top_root, 1 << (two_adicity - lg), p)
    winv = pow(w, p - 2, p)
    ninv = pow(n, p - 2, p)
    c = [sum(x[k] * pow(winv, j * k, p) for k in range(n)) * ninv % p for j in range(n)]
    we = pow(top_root, 1 << (two_adicity - lg - lgb), p)
    out = []
    for k in range(n << lgb):
        pt = gen * pow(we, k, p) % p
        out.append(sum(c[j] * pow(pt, j, p) for j in range(n)) % p)
    return out, c


def make_lde():
    cases = []
    R = 1 << 32
    R256 = 1 << 256
    for lg, lgb in ((0, 2), (1, 1), (3, 1), (3, 2), (4, 3)):
        x = recipe.ntt_input("gl64", lg, 0x5eed5eed0004 + lg)
        y, c = py_lde([int(v) for v in x], lg, lgb, O.GL64_P, 0x185629dcda58878c, 32, 7)
        cases.append({"field": "gl64", "lg": lg, "lg_blowup": lgb, "input": hexs(x),
                      "expect": hexs(np.array(y, dtype=np.uint64)), "aux": hexs(np.array(c, dtype=np.uint64))})
        p = O.BB31_P
        rinv = pow(R, p - 2, p)
        x = recipe.ntt_input("bb31", lg, 0x5eed5eed0004 + lg)
        y, c = py_lde([int(v) * rinv % p for v in x], lg, lgb, p, 0x1ffffedc * rinv % p, 27, 3)
        cases.append({"field": "bb31", "lg": lg, "lg_blowup": lgb, "input": hexs(x),
                      "expect": hexs(np.array([v * R % p for v in y], dtype=np.uint32)),
                      "aux": hexs(np.array([v * R % p for v in c], dtype=np.uint32))})
        for field, curve, gen, S in (("bls12_381", O.BLS12_381, 7, 32), ("bn254", O.BN254, 5, 28), ("bls12_377", O.BLS12_377, 22, 47),
                                 ("pallas", O.PALLAS, 5, 32), ("vesta", O.VESTA, 5, 32)):
            p = O.FR_MODULUS[curve]
            rinv = pow(R256, p - 2, p)
            x = recipe.ntt_input(field, lg, 0x5eed5eed0005 + lg)
            xc = [int.from_bytes(row.tobytes(), "little") * rinv % p for row in x]
            y, c = py_lde(xc, lg, lgb, p, pow(gen, (p - 1) >> S, p), S, gen)
            tow = lambda vals: np.frombuffer(b"".join((v * R256 % p).to_bytes(32, "little") for v in vals), dtype=np.uint64)
            cases.append({"field": field, "lg": lg, "lg_blowup": lgb, "input": hexs(x),
                          "expect": hexs(tow(y)), "aux": hexs(tow(c))})
    json.dump(cases, open(os.path.join(HERE, "lde_golden.json"), "w"), indent=0)
    print("lde cases:", len(cases))


# ---- polynomial primitives (polynomial/*.cuh), definition level -----------------------------
def make_poly():
    """prefix_op (Add / Multiply), evaluate, div_by_x_minus_z (both rotations) on canonical
    Python integers, all four fields, lengths around the tile edges of the GPU kernels."""
    R32, R256 = 1 << 32, 1 << 256
    fields = {
        "gl64": (O.GL64_P, 1, np.uint64, 8),
        "bb31": (O.BB31_P, R32, np.uint32, 4),
        "bls12_381": (O.FR_MODULUS[O.BLS12_381], R256, np.uint64, 32),
        "bn254": (O.FR_MODULUS[O.BN254], R256, np.uint64, 32),
        "bls12_377": (O.FR_MODULUS[O.BLS12_377], R256, np.uint64, 32),
        "pallas": (O.FR_MODULUS[O.PALLAS], R256, np.uint64, 32),
        "vesta": (O.FR_MODULUS[O.VESTA], R256, np.uint64, 32),
    }
    cases = []
    rng = np.random.default_rng(0x901f)
    for field, (p, R, dt, nb) in fields.items():
        rinv = pow(R, -1, p)
        enc = lambda vals: np.frombuffer(b"".join((v * R % p).to_bytes(nb, "little") for v in vals), dtype=dt)
        # 2049 / 1030: one element past a GPU tile (the long wide case for one curve only: fixture size)
        for ln in ((1, 2, 7, 64, 257, 2049) if nb <= 8 else (1, 7, 257, 1030) if field == "bls12_381" else (1, 7, 130)):
            c = [int.from_bytes(rng.bytes(40), "little") % p for _ in range(ln)]
            if ln >= 7:
                c[3] = 0; c[5] = p - 1; c[ln - 1] = 1
            z = int.from_bytes(rng.bytes(40), "little") % p
            xs = [0, 1, p - 1, z, (z * z + 7) % p]
            add, mul, ra, rm = [], [], 0, 1
            for v in c:
                ra = (ra + v) % p; rm = rm * v % p
                add.append(ra); mul.append(rm)
            ev = [sum(v * pow(x, i, p) for i, v in enumerate(c)) % p for x in xs]
            b = c[:]
            for k in range(ln - 2, -1, -1):
                b[k] = (b[k] + z * b[k + 1]) % p
            assert b[0] == sum(v * pow(z, i, p) for i, v in enumerate(c)) % p      # remainder = p(z)
            rot = b[1:] + b[:1]
            cases.append({"field": field, "len": ln, "coeffs": hexs(enc(c)), "z": hexs(enc([z])), "xs": hexs(enc(xs)),
                          "prefix_add": hexs(enc(add)), "prefix_mul": hexs(enc(mul)), "evaluate": hexs(enc(ev)),
                          "div": hexs(enc(b)), "div_rotate": hexs(enc(rot))})
        # z = 0 and z = 1
        c = [int.from_bytes(rng.bytes(40), "little") % p for _ in range(300)]
        for z in (0, 1):
            b = c[:]
            for k in range(298, -1, -1):
                b[k] = (b[k] + z * b[k + 1]) % p
            cases.append({"field": field, "len": 300, "coeffs": hexs(enc(c)), "z": hexs(enc([z])), "div": hexs(enc(b)),
                          "div_rotate": hexs(enc(b[1:] + b[:1]))})
    # Mersenne31 (canonical u32) and the BabyBear quartic extension F_p[x]/(x^4 + 11) (4 Montgomery u32):
    # field types without NTT parameters in the reference; element arithmetic from the definitions
    M31 = (1 << 31) - 1
    BB = O.BB31_P

    class Ext4:                                             # tuples of 4 canonical ints, x^4 = -11
        zero = (0, 0, 0, 0); one = (1, 0, 0, 0)
        @staticmethod
        def add(a, b): return tuple((x + y) % BB for x, y in zip(a, b))
        @staticmethod
        def mul(a, b):
            t = [0] * 7
            for i in range(4):
                for j in range(4):
                    t[i + j] += a[i] * b[j]
            return tuple((t[k] + (-11) * (t[k + 4] if k < 3 else 0)) % BB for k in range(4))
        @staticmethod
        def rand(): return tuple(int.from_bytes(rng.bytes(8), "little") % BB for _ in range(4))
        @staticmethod
        def enc(vals): return np.array([[c * R32 % BB for c in v] for v in vals], dtype=np.uint32).reshape(-1)

    class Prime31:
        zero = 0; one = 1
        @staticmethod
        def add(a, b): return (a + b) % M31
        @staticmethod
        def mul(a, b): return a * b % M31
        @staticmethod
        def rand(): return int.from_bytes(rng.bytes(8), "little") % M31
        @staticmethod
        def enc(vals): return np.array(vals, dtype=np.uint32)

    for field, K in (("m31", Prime31), ("bb31x4", Ext4)):
        for ln in ((1, 7, 257, 2049) if field == "m31" else (1, 7, 257, 1030)):     # one element past a GPU tile
            c = [K.rand() for _ in range(ln)]
            if ln >= 7:
                c[3] = K.zero; c[ln - 1] = K.one
            z = K.rand()
            xs = [K.zero, K.one, z, K.mul(z, z)]
            add, mul, ra, rm = [], [], K.zero, K.one
            for v in c:
                ra = K.add(ra, v); rm = K.mul(rm, v)
                add.append(ra); mul.append(rm)

            def horner(x):
                acc = K.zero
                for v in reversed(c):
                    acc = K.add(v, K.mul(x, acc))
                return acc
            ev = [horner(x) for x in xs]
            b = c[:]
            for k in range(ln - 2, -1, -1):
                b[k] = K.add(b[k], K.mul(z, b[k + 1]))
            assert b[0] == horner(z)
            cases.append({"field": field, "len": ln, "coeffs": hexs(K.enc(c)), "z": hexs(K.enc([z])), "xs": hexs(K.enc(xs)),
                          "prefix_add": hexs(K.enc(add)), "prefix_mul": hexs(K.enc(mul)), "evaluate": hexs(K.enc(ev)),
                          "div": hexs(K.enc(b)), "div_rotate": hexs(K.enc(b[1:] + b[:1]))})
    json.dump(cases, open(os.path.join(HERE, "poly_golden.json"), "w"), indent=0)
    print("poly cases:", len(cases))


if __name__ == "__main__":
    if "--g2-only" in sys.argv:
        make_msm_g2()
    elif "--lde-only" in sys.argv:
        make_lde()
    elif "--poly-only" in sys.argv:
        make_poly()
    else:
        make_msm()
        make_msm_g2()
        make_ntt()
        make_lde()
        make_poly()
