"""GPU parity tests for the MSM hot path, through the C ABI (ctypes).

Bit-exactness is defined on the affine normalisation of the Jacobian result
(what poc/msm-cuda/tests/msm.rs:26-38 compares)."""
import json
import os

import numpy as np
import pytest

import recipe

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CURVES = [(0, "bls12_381"), (1, "bn254"), (4, "bls12_377"), (6, "pallas"), (7, "vesta")]


def P(a):
    return a.ctypes.data


@pytest.mark.parametrize("curve,name", CURVES)
def test_device_field_ops(oracle, libs, curve, name):
    """k_field_op on the GPU vs the oracle, element by element."""
    from sppark_amd import ffi
    O = oracle
    L = ffi.load_devtest(name)
    rng = np.random.default_rng(curve + 10)
    for which, p, nb, ofield in ((0, O.FP_MODULUS[curve], O.FP_BYTES[curve], O.FP_FIELD_ID[curve]),
                                 (1, O.FR_MODULUS[curve], 32, O.FR_FIELD_ID[curve])):
        n = 512
        va = [int.from_bytes(rng.bytes(nb + 8), "little") % p for _ in range(n)]
        vb = [int.from_bytes(rng.bytes(nb + 8), "little") % p for _ in range(n)]
        va[:5] = [0, 1, p - 1, p - 1, 0]; vb[:5] = [0, p - 1, p - 1, 1, 5]
        a = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in va), dtype=np.uint8).copy()
        b = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in vb), dtype=np.uint8).copy()
        for op, oop in ((0, 0), (1, 1), (2, 2), (3, 7), (4, 6), (5, 5), (6, 4)):
            out = np.zeros_like(a)
            ffi.check(L, L.sppark_devtest_field_op(which, op, P(out), P(a), P(b), n))
            for i in range(n):
                e = O.field_op(ofield, oop, a[i * nb:(i + 1) * nb].view(np.uint64), b[i * nb:(i + 1) * nb].view(np.uint64))
                assert (e.view(np.uint8) == out[i * nb:(i + 1) * nb]).all(), (name, which, op, i)


@pytest.mark.parametrize("curve,name", CURVES)
def test_device_point_ops(oracle, libs, curve, name):
    """xyzz add / mixed add / mixed sub / double on the GPU vs the oracle (affine compare),
    including equal, opposite and infinite operands."""
    from sppark_amd import ffi
    O = oracle
    L = ffi.load_devtest(name)
    fb = O.FP_BYTES[curve]
    n = 64
    A = O.g1_gen_points(curve, n, 1); B = O.g1_gen_points(curve, n, 2)
    B[0] = A[0]                                              # equal -> doubling branch
    pmod = O.FP_MODULUS[curve]
    y = int.from_bytes(A[1, fb:].tobytes(), "little")
    B[1] = A[1]; B[1, fb:] = np.frombuffer(((pmod - y) % pmod).to_bytes(fb, "little"), dtype=np.uint8)   # opposite
    B[2] = 0                                                 # infinity operand
    one = O.field_op(O.FP_FIELD_ID[curve], 4, O.int_to_limbs(1, fb)).view(np.uint8)

    def to_xyzz(aff):
        x = np.zeros((aff.shape[0], 4 * fb), dtype=np.uint8)
        x[:, :2 * fb] = aff; x[:, 2 * fb:3 * fb] = one; x[:, 3 * fb:] = one
        inf = (aff == 0).all(axis=1)
        x[inf] = 0
        return x

    def to_jac(aff):
        j = np.zeros((aff.shape[0], 3 * fb), dtype=np.uint8)
        j[:, :2 * fb] = aff; j[:, 2 * fb:] = one
        j[(aff == 0).all(axis=1)] = 0
        return j

    xa, xb = to_xyzz(A), to_xyzz(B)
    xa[3] = 0                                                # accumulator at infinity
    A3 = A.copy(); A3[3] = 0
    ja, jb = to_jac(A3), to_jac(B)
    negB = B.copy()
    for i in range(n):
        yy = int.from_bytes(B[i, fb:].tobytes(), "little")
        negB[i, fb:] = np.frombuffer(((pmod - yy) % pmod).to_bytes(fb, "little"), dtype=np.uint8)
    negB[2] = 0
    jnb = to_jac(negB)
    for op, operand in ((0, xb), (1, B), (2, B), (3, None)):
        out = np.zeros_like(xa)
        ffi.check(L, L.sppark_devtest_xyzz_op(op, P(out), P(xa), P(operand) if operand is not None else 0, n))
        for i in range(n):
            if op in (0, 1):
                e = O.jac_add(curve, ja[i], jb[i])
            elif op == 2:
                e = O.jac_add(curve, ja[i], jnb[i])
            else:
                e = O.jac_dbl(curve, ja[i])
            assert (O.xyzz_to_affine(curve, out[i]) == O.jac_to_affine(curve, e)).all(), (name, op, i)


def test_bucket_field_ops(libs):
    """ff/montx_dev.hpp (the BLS12-381 bucket pipeline's loosely-reduced 28-bit-limb field),
    element-wise on the GPU against Python big-ints: products with normalised and with fat
    operands, squares, lazy subtraction with fat multiples of p, carry propagation, the exact
    zero test on loosely reduced values, and both conversions from/to the wire form."""
    import random
    from sppark_amd import ffi
    L = ffi.load_devtest("bls12_381")
    NL = L.sppark_devtest_bucket_field_limbs()
    assert NL == 14
    Pm = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    LB = 28; MASK = (1 << LB) - 1; R = 1 << (LB * NL); Rinv = pow(R, Pm - 2, Pm)
    limbs = lambda v: [(v >> (LB * j)) & MASK if j < NL - 1 else v >> (LB * j) for j in range(NL)]
    val = lambda l: sum(int(x) << (LB * j) for j, x in enumerate(l))
    random.seed(7)
    n = 2048

    def rnd(k):
        return random.choice([0, 1, Pm - 1, Pm, Pm + 1, 2 * Pm, k * Pm - 1]) if random.random() < 0.06 else random.randrange(k * Pm)
    A = [rnd(3) for _ in range(n)]; B = [rnd(3) for _ in range(n)]
    a = np.array([limbs(v) for v in A], dtype=np.uint32); b = np.array([limbs(v) for v in B], dtype=np.uint32)
    out = np.zeros_like(a)

    def run(op, a_, b_):
        ffi.check(L, L.sppark_devtest_bucket_field_op(op, P(out), P(a_), P(b_), n))
        return [val(r) for r in out], out.copy()
    got, raw = run(0, a, b)
    assert all(g % Pm == x * y * Rinv % Pm and g <= x * y // R + Pm for g, x, y in zip(got, A, B))
    assert (raw[:, :NL - 1] <= MASK).all()
    got, _ = run(1, a, b)
    assert all(g % Pm == x * x * Rinv % Pm for g, x in zip(got, A))

    def fat(K, Bw):
        pl = limbs(K * Pm)
        return [pl[j] + ((Bw << LB) if j < NL - 1 else 0) - (Bw if j > 0 else 0) for j in range(NL)]
    F = fat(12, 6)
    C = [random.randrange(11 * Pm) for _ in range(n)]
    a_fat = np.array([[x + f - y for x, f, y in zip(limbs(A[i]), F, limbs(C[i]))] for i in range(n)], dtype=np.uint32)
    assert int(a_fat.max()) < (1 << 31)
    AF = [val(r) for r in a_fat]
    got, _ = run(0, a_fat, b)                                  # fat left operand
    assert all(g % Pm == x * y * Rinv % Pm for g, x, y in zip(got, AF, B))
    B2 = [v % (2 * Pm) for v in B]; b2 = np.array([limbs(v) for v in B2], dtype=np.uint32)
    got, raw = run(2, a, b2)
    assert all(g == x + 3 * Pm - y for g, x, y in zip(got, A, B2)) and (raw[:, :NL - 1] <= MASK).all()
    got, raw = run(3, a, b)
    assert all(g == x + y for g, x, y in zip(got, A, B)) and (raw[:, :NL - 1] <= MASK).all()
    Z = [random.choice([0, Pm, 2 * Pm, 5 * Pm, 12 * Pm, 7 * Pm + 1, Pm - 1, random.randrange(13 * Pm),
                        (random.randrange(13 * Pm) & ~MASK) | ((k * Pm) & MASK)]) for k in range(n)]
    z = np.array([limbs(v) for v in Z], dtype=np.uint32)
    _, raw = run(4, z, b)
    assert all(int(raw[i, 0]) == (1 if Z[i] % Pm == 0 else 0) for i in range(n))
    # wire form (x * 2^384, 12 words, canonical) <-> internal (x * 2^392)
    W = [random.choice([0, 1, Pm - 1]) if random.random() < 0.05 else random.randrange(Pm) for _ in range(n)]
    w = np.zeros((n, NL), dtype=np.uint32)
    for i, v in enumerate(W):
        w[i, :12] = [(v >> (32 * k)) & 0xffffffff for k in range(12)]
    got, raw = run(5, w, b)
    assert all(g % Pm == v * 256 % Pm and g < 2 * Pm for g, v in zip(got, W)) and (raw[:, :NL - 1] <= MASK).all()
    _, raw = run(6, a_fat, b)                                  # any admissible lazy value -> canonical wire words
    inv256 = pow(256, Pm - 2, Pm)
    for i in range(n):
        assert sum(int(raw[i, k]) << (32 * k) for k in range(12)) == AF[i] * inv256 % Pm


def test_bucket_point_ops_bit_exact_with_wire_class(oracle, libs):
    """The loosely-reduced XYZZ formulas (ec/xyzzx_dev.hpp), fed and read back in the wire form,
    give the SAME coordinates bit for bit as the canonical 32-bit-limb class (same formulas,
    same field elements), including equal, opposite and infinite operands."""
    from sppark_amd import ffi
    O = oracle
    L = ffi.load_devtest("bls12_381")
    curve, fb, n = 0, 48, 96
    A = O.g1_gen_points(curve, n, 11); B = O.g1_gen_points(curve, n, 12)
    B[0] = A[0]
    pmod = O.FP_MODULUS[curve]
    y = int.from_bytes(A[1, fb:].tobytes(), "little")
    B[1] = A[1]; B[1, fb:] = np.frombuffer(((pmod - y) % pmod).to_bytes(fb, "little"), dtype=np.uint8)
    B[2] = 0
    one = O.field_op(O.FIELD_BLS_FP, 4, O.int_to_limbs(1, fb)).view(np.uint8)
    xa = np.zeros((n, 4 * fb), dtype=np.uint8); xa[:, :2 * fb] = A; xa[:, 2 * fb:3 * fb] = one; xa[:, 3 * fb:] = one
    xa[3] = 0
    # make the accumulators non-trivial (ZZ, ZZZ != 1): a few additions with the wire class first
    tmp = np.zeros_like(xa)
    ffi.check(L, L.sppark_devtest_xyzz_op(1, P(tmp), P(xa), P(O.g1_gen_points(curve, n, 13)), n)); xa = tmp.copy()
    xb = np.zeros_like(xa)
    ffi.check(L, L.sppark_devtest_xyzz_op(1, P(xb), P(xa), P(O.g1_gen_points(curve, n, 14)), n))
    xb[0] = xa[0]                                            # equal XYZZ operands -> doubling inside add
    for op, operand in ((0, xb), (1, B), (2, B), (3, None)):
        ref = np.zeros_like(xa); got = np.zeros_like(xa)
        ptr = P(operand) if operand is not None else 0
        ffi.check(L, L.sppark_devtest_xyzz_op(op, P(ref), P(xa), ptr, n))
        ffi.check(L, L.sppark_devtest_bucket_xyzz_op(op, P(got), P(xa), ptr, n))
        assert (got == ref).all(), op
    # the cooperative forms (ec/xyzz_coop.hpp: four waves per 64 operations), two operations in a row:
    # (a + b) + b against two serial additions, 2(2a) against two serial doublings; n is not a multiple of 64
    xb[5] = 0                                                # an operand at infinity on the right as well
    r1 = np.zeros_like(xa); r2 = np.zeros_like(xa); got = np.zeros_like(xa)
    ffi.check(L, L.sppark_devtest_xyzz_op(0, P(r1), P(xa), P(xb), n)); ffi.check(L, L.sppark_devtest_xyzz_op(0, P(r2), P(r1), P(xb), n))
    ffi.check(L, L.sppark_devtest_bucket_xyzz_op(4, P(got), P(xa), P(xb), n))
    assert (got == r2).all()
    ffi.check(L, L.sppark_devtest_xyzz_op(3, P(r1), P(xa), 0, n)); ffi.check(L, L.sppark_devtest_xyzz_op(3, P(r2), P(r1), 0, n))
    ffi.check(L, L.sppark_devtest_bucket_xyzz_op(5, P(got), P(xa), 0, n))
    assert (got == r2).all()


@pytest.mark.parametrize("curve,name", CURVES)
def test_generate_points_matches_oracle(oracle, libs, curve, name):
    """sppark_g1_generate (device double-and-add + host batch normalisation)."""
    import sppark_amd
    O = oracle
    fb = O.FP_BYTES[curve]
    out = np.zeros((33, 2 * fb), dtype=np.uint8)
    sppark_amd.generate_points(out, 33, 0xabcdef, 2 * fb, name)
    assert (out == O.g1_gen_points(curve, 33, 0xabcdef)).all()
    flagged = np.zeros((5, 2 * fb + 8), dtype=np.uint8)
    sppark_amd.generate_points(flagged, 5, 7, 2 * fb + 8, name)
    assert (flagged[:, :2 * fb] == O.g1_gen_points(curve, 5, 7)).all() and (flagged[:, 2 * fb:] == 0).all()


@pytest.mark.parametrize("curve,name", CURVES)
def test_generate_progression_matches_oracle(oracle, libs, curve, name):
    """sppark_g1_generate_progression: P_i = (a + i b) G, all distinct, normalised on the device.  3 * 2^18 + 5 points (every
    lane walks three or four steps of (T b) G), sampled indices against the oracle's scalar multiplication, plain and
    flagged strides; one lane per point below 2^18; b = 0 (the step is the point at infinity)."""
    import torch
    import sppark_amd
    O = oracle
    fb = O.FP_BYTES[curve]
    G = O.g1_generator(curve)
    a, b = 0x1d2c3b4a59687766554433221100ffee, 0xc0ffee11d00dfeedbeef0123
    n = 3 * (1 << 18) + 5
    rng = np.random.default_rng(curve)
    for stride in (2 * fb, 2 * fb + 8):
        out = torch.full((n, stride), 0xa5, dtype=torch.uint8, device="cuda")
        sppark_amd.generate_progression(out, n, a, b, stride, name)
        idx = sorted(set([0, 1, 63, 64, (1 << 18) - 1, 1 << 18, (1 << 18) + 1, 2 << 18, 3 << 18, n - 1] + list(rng.integers(0, n, 40))))
        got = out[torch.tensor(idx, device="cuda")].cpu().numpy()
        for row, i in zip(got, idx):
            assert (row[:2 * fb] == O.g1_mul(curve, G, a + int(i) * b)).all(), (name, stride, i)
            assert (row[2 * fb:] == 0).all()
    small = torch.zeros((100, 2 * fb), dtype=torch.uint8, device="cuda")
    sppark_amd.generate_progression(small, 100, 5, 3, 2 * fb, name)
    for i in (0, 1, 99):
        assert (small[i].cpu().numpy() == O.g1_mul(curve, G, 5 + 3 * i)).all()
    sppark_amd.generate_progression(small, 100, 7, 0, 2 * fb, name)
    assert (small.cpu().numpy() == O.g1_mul(curve, G, 7)).all()
    with pytest.raises(sppark_amd.SpparkError):
        sppark_amd.generate_progression(small, 100, 0, 3, 2 * fb, name)          # a = 0 would be the point at infinity


def test_msm_golden_vectors(oracle, libs):
    """Golden vectors produced by the reference's own msm/pippenger.hpp."""
    import sppark_amd
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "msm_golden.json"))):
        curve = O.CURVE_ID[c["curve"]]
        fb = O.FP_BYTES[curve]
        stride = 2 * fb + 8 if c["flagged"] else 2 * fb
        if "points" in c:
            pts = np.frombuffer(bytes.fromhex(c["points"]), dtype=np.uint8).reshape(c["n"], stride).copy()
            sc = np.frombuffer(bytes.fromhex(c["scalars"]), dtype=np.uint8).reshape(c["n"], 32).copy()
        else:
            pts, sc = recipe.msm_inputs(curve, c["n"], c["seed"], c["ndistinct"], c["flagged"])
        exp = np.frombuffer(bytes.fromhex(c["expect_affine"]), dtype=np.uint8)
        if c["flagged"]:
            out = sppark_amd.multi_scalar_mult_arkworks(pts, sc, c["curve"])
        else:
            out = sppark_amd.multi_scalar_mult(pts, sc, c["curve"])
        assert (sppark_amd.to_affine(out, c["curve"]) == exp).all(), (c["curve"], c["n"])
        assert (O.jac_to_affine(curve, out) == exp).all()


@pytest.mark.parametrize("curve,name", CURVES)
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 1000, 4097, 1 << 15])
def test_msm_vs_oracle(oracle, libs, curve, name, n):
    """Reference test shape (poc/msm-cuda/tests/msm.rs:19-39, TEST_NPOW=15) +
    ragged sizes, both wire formats."""
    import sppark_amd
    O = oracle
    for flagged in (False, True):
        pts, sc = recipe.msm_inputs(curve, n, 31337 + n, ndistinct=2048, flagged=flagged)
        exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
        f = sppark_amd.multi_scalar_mult_arkworks if flagged else sppark_amd.multi_scalar_mult
        out = f(pts, sc, name)
        assert (sppark_amd.to_affine(out, name) == exp).all(), (name, n, flagged)


def test_msm_empty_and_all_infinity(oracle, libs):
    import sppark_amd
    out = sppark_amd.multi_scalar_mult(np.zeros((0, 96), dtype=np.uint8), np.zeros((0, 32), dtype=np.uint8))
    assert (out == 0).all()
    pts = np.zeros((100, 96), dtype=np.uint8)
    sc = oracle.random_scalars(0, 100, 3)
    assert (sppark_amd.to_affine(sppark_amd.multi_scalar_mult(pts, sc)) == 0).all()
    pts2, _ = recipe.msm_inputs(0, 100, 9)
    assert (sppark_amd.to_affine(sppark_amd.multi_scalar_mult(pts2, np.zeros((100, 32), dtype=np.uint8))) == 0).all()


@pytest.mark.parametrize("tune", [dict(wbits=7, L=4, F=4, K=2, nslabs=3), dict(wbits=12, L=16, F=8, K=4, nslabs=2),
                                  dict(wbits=16, L=64, F=32, K=8, nslabs=1), dict(wbits=2, L=4, F=4, K=2, nslabs=1),
                                  dict(wbits=20, L=8, F=8, K=8, nslabs=2), dict(wbits=23, L=4, F=16, K=16, nslabs=1)])
def test_msm_tunables(oracle, libs, tune):
    """every plan parameter away from its default must give the same group element"""
    import sppark_amd
    O = oracle
    ctx = sppark_amd.MsmContext("bls12_381")
    ctx.tune(**tune)
    pts, sc = recipe.msm_inputs(0, 3000, 17, ndistinct=128)
    out = ctx.invoke(pts, sc)
    assert (sppark_amd.to_affine(out) == O.msm_affine(0, pts, sc, algo=0, param=8)).all()
    ctx.close()


@pytest.mark.parametrize("curve,name", [(0, "bls12_381"), (1, "bn254")])
def test_msm_tail_variants(oracle, libs, curve, name):
    """The tail of an MSM (record list + bucket sums) in every form the driver can run it: k_join_runs on / off,
    with and without the one-launch narrow end of the tree (k_reduce_tail) and the low-latency bucket-sum kernels,
    the first level's two sums on one lane instead of two waves (10), first-level chunks of 4 / 16 -- on inputs where the join resolves everything (many buckets, short segments), where
    it resolves nothing (one bucket holds all entries: the tree does the work) and in between."""
    import sppark_amd
    O = oracle
    ctx = sppark_amd.MsmContext(name)
    n = 40000
    pts, sc = recipe.msm_inputs(curve, n, 23, ndistinct=600, edge=False)
    s_eq = sc.copy(); s_eq[:] = sc[1]
    s_two = sc.copy(); s_two[: n // 3] = sc[0]; s_two[n // 3:] = sc[1]
    for scal, plans in ((sc, (dict(wbits=14, L=8), dict(wbits=10, L=16), dict())), (s_eq, (dict(wbits=12, L=8), dict())),
                        (s_two, (dict(wbits=9, L=4, F=4),))):
        exp = O.msm_affine(curve, pts, scal, algo=0, param=8)
        for plan in plans:
            for join, k1 in ((0, 0), (1, 0), (2, 0), (3, 0), (10, 0), (0, 4), (0, 16)):
                ctx.tune(**plan); ctx.tune_tail(join, k1)
                out = ctx.invoke(pts, scal)
                assert (sppark_amd.to_affine(out, name) == exp).all(), (plan, join, k1)
    ctx.close()


@pytest.mark.parametrize("curve,name", CURVES)
def test_msm_piece_tree_of_small_sizes(oracle, libs, curve, name):
    """Small MSMs (up to 2^16 points: a few buckets of hundreds of entries per window) sum the pieces of a bucket by a tree over
    the bucket's own pieces (msm_piece_kernels.hpp), sized for the average bucket; the fan-in tree is run AFTERWARDS, and the
    bucket sums again, only when the device reports a bucket beyond that size.  Uniform scalars at the automatic plan and at
    forced ones (aligned / unaligned bucket starts, odd piece counts, ragged sizes): against the oracle and WITHOUT a second
    pass; all scalars equal, a mix, 16-bit scalars, every second scalar zero: against the oracle, the heavy buckets through the
    second pass; the same calls with the piece tree switched off (tune_tail 5), without the cooperative kernels (4), with a
    launch per level instead of the one-launch narrow end k_piece_tail_coop (8), and with that launch taking every level /
    only the last ones (16 + 30, 16 + 10: work-groups of 1 ... 64 buckets)."""
    import sppark_amd
    O = oracle
    ctx = sppark_amd.MsmContext(name)
    for n, plans in ((1 << 12, (dict(), dict(wbits=5, L=7))), (5000, (dict(), dict(wbits=6, L=16))), ((1 << 15) + 17, (dict(),)),
                     (1 << 16, (dict(), dict(wbits=9, L=8)))):
        pts, sc = recipe.msm_inputs(curve, n, 31 + n, ndistinct=700, flagged=True, edge=True)
        s_eq = sc.copy(); s_eq[:] = sc[2]
        s_mix = sc.copy(); s_mix[n // 5:] = sc[1]
        s_16 = np.zeros_like(sc); s_16[:, :2] = sc[:, :2]
        s_half = sc.copy(); s_half[::2] = 0
        for what, s_ in (("uniform", sc), ("equal", s_eq), ("mix", s_mix), ("16-bit", s_16), ("half zero", s_half)):
            if n > 5000 and what in ("16-bit", "half zero") and curve != 0:
                continue
            exp = O.msm_affine(curve, pts, s_, algo=0, param=8)
            for plan in plans:
                for join in (0, 5, 4, 8, 46, 26):
                    ctx.tune(**plan); ctx.tune_tail(join, 0)
                    before = ctx.tail_redone()
                    out = ctx.invoke(pts, s_, ffi_affine_sz=pts.shape[1])
                    assert (sppark_amd.to_affine(out, name) == exp).all(), (n, what, plan, join)
                    redone = ctx.tail_redone() - before
                    if join == 5 or what == "uniform":
                        assert redone == 0, (n, what, plan, join)
                    pl = ctx.plan(n)
                    if join != 5 and what == "equal" and n // pl["run_length"] > pl["piece_tree_max"] + 1:
                        assert redone == 1, (n, what, plan, join)      # ONE bucket per window holds everything: beyond the tree's size
    ctx.close()


def test_msm_skewed_scalars(oracle, libs):
    """SURVEY 8(d) skew cases: all scalars equal, 50% zeros, 16-bit scalars,
    all points equal."""
    import sppark_amd
    O = oracle
    n = 20000
    pts, sc = recipe.msm_inputs(0, n, 5, ndistinct=512, edge=False)
    s_eq = sc.copy(); s_eq[:] = sc[0]
    s_half = sc.copy(); s_half[::2] = 0
    s_16 = np.zeros_like(sc); s_16[:, :2] = sc[:, :2]
    same = pts.copy(); same[:] = pts[0]
    for p, s in ((pts, s_eq), (pts, s_half), (pts, s_16), (same, sc), (same, s_eq)):
        out = sppark_amd.multi_scalar_mult(p, s)
        assert (sppark_amd.to_affine(out) == O.msm_affine(0, p, s, algo=0, param=8)).all()


def test_msm_montgomery_scalars_and_device_pointers(oracle, libs):
    import torch
    import sppark_amd
    O = oracle
    n = 5000
    pts, sc = recipe.msm_inputs(1, n, 8, ndistinct=256, flagged=True)
    exp = O.msm_affine(1, pts, sc, algo=0, param=8)
    mont = np.zeros_like(sc)
    for i in range(n):
        mont[i] = O.field_op(O.FIELD_BN_FR, 4, sc[i].view(np.uint64)).view(np.uint8)
    ctx = sppark_amd.MsmContext("bn254")
    assert (sppark_amd.to_affine(ctx.invoke(pts, mont, mont=True, ffi_affine_sz=72), "bn254") == exp).all()
    d_pts = torch.from_numpy(pts).cuda(); d_sc = torch.from_numpy(sc).cuda()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert (sppark_amd.to_affine(ctx.invoke(d_pts, d_sc, ffi_affine_sz=72), "bn254") == exp).all()
    assert (sppark_amd.to_affine(ctx.invoke(d_pts, sc, ffi_affine_sz=72), "bn254") == exp).all()
    ctx.close()


def test_msm_preloaded_points(oracle, libs):
    """msm_t(points, np, ffi_affine_sz) + invoke(out, scalars) (pippenger.cuh:351-385,604-605):
    bases copied once into HBM, then MSMs over the first n <= np of them."""
    import torch
    import sppark_amd
    from sppark_amd import ffi
    O = oracle
    n = 3000
    pts, sc = recipe.msm_inputs(0, n, 21, ndistinct=300, flagged=True)
    ctx = sppark_amd.MsmContext("bls12_381")
    with pytest.raises(ffi.SpparkError):                        # nothing preloaded yet
        ctx.invoke(None, sc, ffi_affine_sz=104)
    ctx.set_points(pts, ffi_affine_sz=104)
    assert ctx.preloaded() == n
    assert (sppark_amd.to_affine(ctx.invoke(None, sc)) == O.msm_affine(0, pts, sc, algo=0, param=8)).all()
    sc2 = sc[::-1].copy()
    assert (sppark_amd.to_affine(ctx.invoke(None, sc2)) == O.msm_affine(0, pts, sc2, algo=0, param=8)).all()
    m = 1000                                                    # a prefix, device-resident scalars
    d_sc = torch.from_numpy(sc[:m].copy()).cuda()
    assert (sppark_amd.to_affine(ctx.invoke(None, d_sc)) == O.msm_affine(0, pts[:m], sc[:m], algo=0, param=8)).all()
    with pytest.raises(ffi.SpparkError):                        # more scalars than preloaded points
        ctx.invoke(None, np.concatenate([sc, sc]))
    # replace the set from a DEVICE buffer in the unflagged 96-byte layout
    pts96 = np.ascontiguousarray(pts[:, :96])
    inf = pts[:, 96] != 0
    pts96[inf] = 0
    ctx.set_points(torch.from_numpy(pts96).cuda(), ffi_affine_sz=96)
    assert (sppark_amd.to_affine(ctx.invoke(None, sc)) == O.msm_affine(0, pts, sc, algo=0, param=8)).all()
    ctx.set_points(None)
    assert ctx.preloaded() == 0
    ctx.close()


@pytest.mark.parametrize("name,curve", [("bls12_381", 0), ("bn254", 1), ("bls12_377", 4), ("pallas", 6)])
def test_msm_fixed_base_tables(oracle, libs, name, curve):
    """sppark_msm_set_points_fixed_base: the per-window multiples of the preloaded bases, and an MSM over all of
    them as ONE window over windows x npoints (digit, multiple) pairs -- against the oracle, for host and device
    scalars, Montgomery-form scalars, edge-case inputs (infinity, zero and r - 1 scalars, repeated points), forced
    window widths (many narrow windows ... fewer, wider ones than the size asks for), and the fall-back to the plain
    path for a prefix."""
    import torch
    import sppark_amd
    from sppark_amd import ffi
    O = oracle
    ctx = sppark_amd.MsmContext(name)
    # (below 2^23 points the tables are built only when a width is forced: include/sppark_amd.h)
    for n, wb, flagged in ((1, 8, False), (300, 10, True), (5000, 13, False), (5000, 9, True), (3000, 21, False), (700, 26, False),
                           ((1 << 16) + 3, 16, True), (5000, 0, True)):
        pts, sc = recipe.msm_inputs(curve, n, 900 + n + wb, ndistinct=min(n, 700), flagged=flagged)
        ctx.tune(wbits=wb)
        ctx.set_points(pts, ffi_affine_sz=pts.shape[1], fixed_base=True)
        ctx.tune(wbits=0)                                   # the tables keep the width they were built with
        W = ctx.fixed_base_windows()
        assert ctx.preloaded() == n
        assert W == (-(-O.FR_MODULUS[curve].bit_length() // wb) if wb else 0)
        exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
        assert (sppark_amd.to_affine(ctx.invoke(None, sc), name) == exp).all(), (name, n, wb, "host scalars")
        d_sc = torch.from_numpy(sc).cuda()
        assert (sppark_amd.to_affine(ctx.invoke(None, d_sc), name) == exp).all(), (name, n, wb, "device scalars")
        assert (sppark_amd.to_affine(ctx.invoke(None, d_sc), name) == exp).all(), (name, n, wb, "again")
        if n >= 300:                                        # a prefix: the plain path on the points themselves
            m = n // 3
            assert (sppark_amd.to_affine(ctx.invoke(None, d_sc[:m].contiguous(), npoints=m), name)
                    == O.msm_affine(curve, pts[:m], sc[:m], algo=0, param=8)).all(), (name, n, wb, "prefix")
    # a refused call (stride below two coordinates) leaves the context as it was
    with pytest.raises(ffi.SpparkError):
        ctx.set_points(pts, ffi_affine_sz=8, fixed_base=True)
    assert ctx.preloaded() == n
    assert (sppark_amd.to_affine(ctx.invoke(None, sc), name) == exp).all()
    # plain preload afterwards drops the tables
    ctx.set_points(pts, ffi_affine_sz=pts.shape[1])
    assert ctx.fixed_base_windows() == 0
    assert (sppark_amd.to_affine(ctx.invoke(None, sc), name) == exp).all()
    ctx.set_points(None)
    assert ctx.preloaded() == 0 and ctx.fixed_base_windows() == 0
    ctx.close()


def test_msm_large_linearity(oracle, libs):
    """2^20 points: MSM(P, a) + MSM(P, b) == MSM(P, a+b mod r) and the 2^16
    prefix equals the oracle -- size-independent properties at a size the
    oracle cannot check directly in seconds."""
    import torch
    import sppark_amd
    O = oracle
    n = 1 << 20
    r = O.FR_MODULUS[0]
    base = np.zeros((2048, 96), dtype=np.uint8)
    sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
    pts = base[np.arange(n) % 2048].copy()
    pts[3] = 0
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x3f
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); b[:, 31] &= 0x3f
    a64 = a.view(np.uint64).reshape(n, 4); b64 = b.view(np.uint64).reshape(n, 4)
    s = np.zeros_like(a64); carry = np.zeros(n, dtype=np.uint64)
    for k in range(4):                                       # a + b < 2^255 < 2r: one conditional subtraction
        t = a64[:, k] + b64[:, k]; c1 = t < a64[:, k]
        t2 = t + carry; c2 = t2 < t
        s[:, k] = t2; carry = (c1 | c2).astype(np.uint64)
    s_int_hi = s[:, 3]
    r64 = np.frombuffer(r.to_bytes(32, "little"), dtype=np.uint64)
    ge = np.zeros(n, dtype=bool); decided = np.zeros(n, dtype=bool)
    for k in (3, 2, 1, 0):
        gt = (s[:, k] > r64[k]) & ~decided; lt = (s[:, k] < r64[k]) & ~decided
        ge |= gt; decided |= gt | lt
    ge |= ~decided
    borrow = np.zeros(n, dtype=np.uint64)
    for k in range(4):
        sub = np.where(ge, r64[k], np.uint64(0))
        t = s[:, k] - sub; b1 = s[:, k] < sub
        t2 = t - borrow; b2 = t < borrow
        s[:, k] = t2; borrow = (b1 | b2).astype(np.uint64)
    sc_sum = s.view(np.uint8).reshape(n, 32)
    ctx = sppark_amd.MsmContext("bls12_381")
    d_pts = torch.from_numpy(pts).cuda()
    ra = ctx.invoke(d_pts, a); rb = ctx.invoke(d_pts, b); rs = ctx.invoke(d_pts, sc_sum)
    assert (sppark_amd.to_affine(sppark_amd.jacobian_sum(np.stack([ra, rb]))) == sppark_amd.to_affine(rs)).all()
    m = 1 << 16
    assert (sppark_amd.to_affine(ctx.invoke(pts[:m], a[:m])) == O.msm_affine(0, pts[:m], a[:m], algo=0, param=8)).all()
    ctx.close()


def test_go_bridge_smoke_and_gpu_ptr(libs):
    """poc/go/poc_test.go:8-15: cuda_func runs when IsCudaAvailable(); gpu_ptr_t
    clone/drop keep the ref-count protocol of util/gpu_t.cuh:269-318."""
    import ctypes
    from sppark_amd import ffi
    L = ffi.load("bls12_381")
    assert L.cuda_available()
    err = L.cuda_func(None)
    assert err.code == 0 and ctypes.string_at(err.message)
    L.drop_error_message(err.message)
    L.sppark_gpu_ptr_alloc.argtypes = [ctypes.c_size_t]; L.sppark_gpu_ptr_alloc.restype = ctypes.c_void_p
    L.sppark_gpu_ptr_get.argtypes = [ctypes.POINTER(ctypes.c_void_p)]; L.sppark_gpu_ptr_get.restype = ctypes.c_void_p
    a = ctypes.c_void_p(L.sppark_gpu_ptr_alloc(1 << 20))
    assert a.value and L.sppark_gpu_ptr_get(ctypes.byref(a))
    b = ctypes.c_void_p(L.clone_gpu_ptr_t(ctypes.byref(a)))
    assert b.value == a.value
    L.drop_gpu_ptr_t(ctypes.byref(a))
    assert a.value is None and L.sppark_gpu_ptr_get(ctypes.byref(b))      # still alive through the clone
    L.drop_gpu_ptr_t(ctypes.byref(b))


# ------------------------------------------------------------------- G2 --------
G2 = [("bls12_381", 2), ("bn254", 3), ("bls12_377", 5)]
G2_PATHS = [(1, "wave_pairs"), (2, "one_lane")]


@pytest.fixture(params=G2_PATHS, ids=[p[1] for p in G2_PATHS])
def g2_path(request, libs):
    """Both accumulation kernels of the G2 entry point (sppark_msm_g2_path): a pair of waves per addition with one Fp2
    component each (msm_g2c_kernels.hpp; the default over the 14-limb base fields) and one lane per addition
    (k_accumulate<fp2x_dev>; the default over alt_bn128).  Every G2 test runs under both, on every curve."""
    import sppark_amd
    for name, _ in G2:
        sppark_amd.set_g2_path(request.param[0], name)
    yield request.param[1]
    for name, _ in G2:
        sppark_amd.set_g2_path(0, name)


def test_msm_g2_golden_vectors(oracle, libs, g2_path):
    """mult_pippenger_fp2_inf against the vectors produced by the reference's own templates
    over Fp2 and the 30*G2 KAT of an independent Python group law."""
    import sppark_amd
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "msm_g2_golden.json"))):
        curve = O.CURVE_ID_G2[c["curve"]]
        fb = O.FP_BYTES[curve]
        stride = 2 * fb + 8 if c["flagged"] else 2 * fb
        if "points" in c:
            pts = np.frombuffer(bytes.fromhex(c["points"]), dtype=np.uint8).reshape(c["n"], stride).copy()
            sc = np.frombuffer(bytes.fromhex(c["scalars"]), dtype=np.uint8).reshape(c["n"], 32).copy()
        else:
            pts, sc = recipe.msm_inputs(curve, c["n"], c["seed"], c["ndistinct"], c["flagged"])
        exp = np.frombuffer(bytes.fromhex(c["expect_affine"]), dtype=np.uint8)
        out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, c["curve"], ffi_affine_sz=stride)
        assert (sppark_amd.to_affine_g2(out, c["curve"]) == exp).all(), (c["curve"], c["n"])
        assert (O.jac_to_affine(curve, out) == exp).all()


@pytest.mark.parametrize("name,curve", G2)
@pytest.mark.parametrize("n", [1, 3, 64, 65, 1000, 4097, 1 << 14])
def test_msm_g2_vs_oracle(oracle, libs, g2_path, name, curve, n):
    """the shape of poc/msm-cuda/tests/msm.rs:41-63 (G2 against a CPU MSM) + ragged sizes"""
    import sppark_amd
    O = oracle
    pts, sc = recipe.msm_inputs(curve, n, 4242 + n, ndistinct=512, flagged=True)
    out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name)
    assert (sppark_amd.to_affine_g2(out, name) == O.msm_affine(curve, pts, sc, algo=0, param=8)).all(), (name, n)


def test_msm_g2_edge_cases(oracle, libs, g2_path):
    import torch
    import sppark_amd
    O = oracle
    curve, name = O.BLS12_381_G2, "bls12_381"
    n = 2000
    pts, sc = recipe.msm_inputs(curve, n, 606, ndistinct=128, flagged=True)
    # empty / all infinity / all-zero scalars
    assert (sppark_amd.multi_scalar_mult_fp2_arkworks(pts[:0], sc[:0], name)[192:] == 0).all()
    inf = pts.copy(); inf[:, 192] = 1
    assert (sppark_amd.multi_scalar_mult_fp2_arkworks(inf, sc, name)[192:] == 0).all()
    assert (sppark_amd.multi_scalar_mult_fp2_arkworks(pts, np.zeros_like(sc), name)[192:] == 0).all()
    # skew: all scalars equal, all points equal (doubling path), device-resident inputs
    s_eq = sc.copy(); s_eq[:] = sc[0]
    same = pts.copy(); same[:] = pts[0]
    for p_, s_ in ((pts, s_eq), (same, sc), (same, s_eq)):
        out = sppark_amd.multi_scalar_mult_fp2_arkworks(p_, s_, name)
        assert (sppark_amd.to_affine_g2(out, name) == O.msm_affine(curve, p_, s_, algo=0, param=8)).all()
    out = sppark_amd.multi_scalar_mult_fp2_arkworks(torch.from_numpy(pts).cuda(), torch.from_numpy(sc).cuda(), name)
    exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
    assert (sppark_amd.to_affine_g2(out, name) == exp).all()
    # host combine helper: sum of two halves == whole
    a = sppark_amd.multi_scalar_mult_fp2_arkworks(pts[:n // 2], sc[:n // 2], name)
    b = sppark_amd.multi_scalar_mult_fp2_arkworks(pts[n // 2:], sc[n // 2:], name)
    assert (sppark_amd.to_affine_g2(sppark_amd.jacobian_sum_g2(np.stack([a, b]), name), name) == exp).all()


def test_msm_g2_large_linearity(oracle, libs, g2_path):
    """2^18 G2 points: MSM(P, a) + MSM(P, b) == MSM(P, a + b mod r); the 2^12 prefix equals the oracle."""
    import sppark_amd
    O = oracle
    curve, name = O.BN254_G2, "bn254"
    n = 1 << 18
    base, sc = recipe.msm_inputs(curve, 1 << 12, 99, ndistinct=1024, flagged=True, edge=False)
    pts = base[np.arange(n) % base.shape[0]]
    rng = np.random.default_rng(17)
    r = O.FR_MODULUS[curve]
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x1f
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); b[:, 31] &= 0x0f
    ai = a.view(np.uint64).astype(object); bi = b.view(np.uint64).astype(object)
    c = np.zeros_like(a)
    for i in range(n):
        va = sum(int(ai[i, k]) << (64 * k) for k in range(4)); vb = sum(int(bi[i, k]) << (64 * k) for k in range(4))
        c[i] = np.frombuffer(((va + vb) % r).to_bytes(32, "little"), dtype=np.uint8)
    A = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, a, name)
    B = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, b, name)
    C = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, c, name)
    assert (sppark_amd.to_affine_g2(sppark_amd.jacobian_sum_g2(np.stack([A, B]), name), name) == sppark_amd.to_affine_g2(C, name)).all()
    m = 1 << 12
    out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts[:m], a[:m], name)
    assert (sppark_amd.to_affine_g2(out, name) == O.msm_affine(curve, pts[:m], a[:m], algo=0, param=8)).all()


def test_concurrent_contexts_and_ntt(oracle, libs):
    """Calls are synchronous but independent contexts may run concurrently (the reference is NOT
    safe for concurrent MSMs on one GPU: device-global work counters, SURVEY 8(b) threading note;
    here every context owns its scratch and stream).  Three host threads: two MSM contexts of
    different curves and an NTT, repeated; every result must equal the oracle's."""
    import threading
    import sppark_amd
    from sppark_amd import NTTInputOutputOrder as Ord
    O = oracle
    jobs = []
    for curve, name, n, seed in ((O.BLS12_381, "bls12_381", 6000, 1), (O.BLS12_381, "bls12_381", 3000, 2), (O.BN254, "bn254", 5000, 3)):
        pts, sc = recipe.msm_inputs(curve, n, 900 + seed, ndistinct=256)
        jobs.append((curve, name, pts, sc, O.msm_affine(curve, pts, sc, algo=0, param=8)))
    x = recipe.ntt_input("gl64", 16, 5)
    x_exp = O.ntt_gl64(x, O.NR, O.FORWARD, O.STANDARD)
    errors = []

    def msm_worker(curve, name, pts, sc, exp):
        try:
            ctx = sppark_amd.MsmContext(name)
            for _ in range(6):
                if not (sppark_amd.to_affine(ctx.invoke(pts, sc), name) == exp).all():
                    errors.append(("msm", name))
            ctx.close()
        except Exception as e:                      # noqa: BLE001
            errors.append(("msm", name, repr(e)))

    def ntt_worker():
        try:
            for _ in range(20):
                if not (sppark_amd.NTT(0, x.copy(), Ord.NR, "gl64") == x_exp).all():
                    errors.append(("ntt",))
        except Exception as e:                      # noqa: BLE001
            errors.append(("ntt", repr(e)))

    threads = [threading.Thread(target=msm_worker, args=j) for j in jobs] + [threading.Thread(target=ntt_worker)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors


def test_msm_randomised_plans_and_inputs(oracle, libs):
    """Randomised stress of the whole pipeline (sort split, run length, fan-in, bucket chunking,
    window width) on inputs with repeated points, repeated scalars, zeros, r-1 and infinities:
    every result must equal the oracle's.  Exercises the loosely-reduced field's bounds on many
    different addition orders."""
    import sppark_amd
    O = oracle
    rng = np.random.default_rng(2024)
    ctxs = {name: sppark_amd.MsmContext(name) for _, name in CURVES}
    for it in range(60):
        curve, name = CURVES[it % 2]
        n = int(rng.choice([1, 2, 5, 33, 100, 257, 1000, 3000, 6000]))
        nd = int(rng.choice([1, 2, 7, 64, 512]))
        flagged = bool(rng.integers(0, 2))
        pts, sc = recipe.msm_inputs(curve, n, 5000 + it, ndistinct=nd, flagged=flagged, edge=bool(rng.integers(0, 2)))
        mode = int(rng.integers(0, 5))
        if mode == 1: sc[:] = sc[0]
        elif mode == 2: sc[rng.integers(0, 2, size=n).astype(bool)] = 0
        elif mode == 3: sc[:, 2:] = 0
        elif mode == 4 and n > 4: sc[: n // 2] = sc[n // 2: 2 * (n // 2)]
        ctx = ctxs[name]
        wb = int(rng.choice([0, 0, 3, 6, 9, 13, 17, 21]))
        ctx.tune(wbits=wb, L=int(rng.choice([0, 4, 16, 64])), F=int(rng.choice([0, 4, 8, 32])),
                 K=int(rng.choice([0, 2, 8])), nslabs=int(rng.choice([0, 1, 3])))
        ctx.tune_sort(int(rng.choice([0, 0, 1, 4])))
        out = ctx.invoke(pts, sc, ffi_affine_sz=pts.shape[1])
        exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
        assert (sppark_amd.to_affine(out, name) == exp).all(), (it, name, n, nd, flagged, mode, wb)
    for c in ctxs.values():
        c.close()


def test_one_shot_entry_points_reuse_and_release_scratch(oracle, libs):
    """mult_pippenger_inf / mult_pippenger_fp2_inf keep one context per (thread, device) between
    calls; results stay right across sizes, curves' groups and after sppark_msm_release_cached()."""
    import sppark_amd
    from sppark_amd import ffi
    O = oracle
    L = ffi.load("bls12_381")
    for n in (3000, 50, 7000):
        pts, sc = recipe.msm_inputs(O.BLS12_381, n, 77 + n, flagged=True)
        assert (sppark_amd.to_affine(sppark_amd.multi_scalar_mult_arkworks(pts, sc)) == O.msm_affine(O.BLS12_381, pts, sc, algo=0, param=8)).all()
        p2, s2 = recipe.msm_inputs(O.BLS12_381_G2, n // 10 + 1, 78 + n, flagged=True)
        assert (sppark_amd.to_affine_g2(sppark_amd.multi_scalar_mult_fp2_arkworks(p2, s2)) == O.msm_affine(O.BLS12_381_G2, p2, s2, algo=0, param=8)).all()
        if n == 50:
            L.sppark_msm_release_cached()
    L.sppark_msm_release_cached()


def test_msm_oversized_sort_partitions(oracle, libs):
    """Skewed scalars put a whole window into one sort partition; partitions above the split
    threshold are sorted cooperatively by several work-groups (k_big_*).  Forced here with a tiny
    threshold on all the skew shapes, against the oracle."""
    import sppark_amd
    O = oracle
    n = 6000
    pts, sc = recipe.msm_inputs(O.BLS12_381, n, 4040, ndistinct=300, flagged=True)
    s_eq = sc.copy(); s_eq[:] = sc[0]
    s_one = np.zeros_like(sc); s_one[:, 0] = 1
    s_small = np.zeros_like(sc); s_small[:, 0] = sc[:, 0] & 7
    s_half = sc.copy(); s_half[::2] = 0
    ctx = sppark_amd.MsmContext("bls12_381")
    for big in (1, 37, 500):
        ctx.tune_split(big)
        for wb, lb in ((0, 0), (13, 3), (9, 0), (17, 6)):
            ctx.tune(wbits=wb); ctx.tune_sort(lb)
            for s in (sc, s_eq, s_one, s_small, s_half):
                out = ctx.invoke(pts, s, ffi_affine_sz=104)
                assert (sppark_amd.to_affine(out) == O.msm_affine(O.BLS12_381, pts, s, algo=0, param=8)).all(), (big, wb, lb)
    ctx.close()


def test_msm_sort_partition_paths(oracle, libs):
    """k_sortB's three ways through a level-A partition in one MSM: up to 18432 entries it is held in
    registers and placed through an LDS image, above that it takes the two-pass path, above the split
    threshold the cooperative kernels -- forced with scalars that are equal for most points (every window
    then has one partition with ~n entries) plus a uniform remainder, for long and short windows."""
    import sppark_amd
    O = oracle
    n = 60000
    pts, sc = recipe.msm_inputs(O.BLS12_381, n, 6060, ndistinct=400, flagged=False)
    s_mix = sc.copy(); s_mix[: n - 9000] = sc[0]                # 51000 equal scalars + 9000 uniform ones
    s_eq = sc.copy(); s_eq[:] = sc[1]
    ctx = sppark_amd.MsmContext("bls12_381")
    for big in (0, 30000):                                      # 0: automatic threshold (2^18): two-pass path
        ctx.tune_split(big)
        for wb, lb in ((0, 0), (17, 6), (21, 9), (22, 9), (13, 0)):
            ctx.tune(wbits=wb); ctx.tune_sort(lb)
            for s_ in (s_mix, s_eq):
                out = ctx.invoke(pts, s_, ffi_affine_sz=96)
                assert (sppark_amd.to_affine(out) == O.msm_affine(O.BLS12_381, pts, s_, algo=0, param=8)).all(), (big, wb, lb)
    ctx.close()


def test_msm_packed_sort_records_index_groups(oracle, libs):
    """4-byte level-A sort records (csrc/msm/msm_sort_records.hpp): the index bits above IB = 31 - LB are not stored, level B
    finds them from the position of a record in its partition.  With the widest k_lo (LB = 13: 18 index bits) 600 000
    points are three index groups, in every form of level B: the register path (2^8 partitions of ~2300 entries), the
    two-pass path (2^3 partitions of ~75 000), the cooperative kernels (most scalars equal: one partition holds nearly the
    whole window; staged slices and, with the threshold at 30 000, direct ones) -- against the oracle, and the same calls
    with 8-byte records (an explicit slab count keeps them)."""
    import sppark_amd
    O = oracle
    n = 600000
    pts, sc = recipe.msm_inputs(O.BLS12_381, n, 7171, ndistinct=500, flagged=False)
    s_mix = sc.copy(); s_mix[3000: n - 40000] = sc[0]           # uniform at both ends, 557 000 equal scalars between
    s_half = sc.copy(); s_half[::2] = 0                         # zero digits: records thin out, the groups' first positions move
    ctx = sppark_amd.MsmContext("bls12_381")
    for s_ in (sc, s_mix, s_half):
        exp = O.msm_affine(O.BLS12_381, pts, s_, algo=0, param=8)
        for wb, lb, big in ((22, 13, 0), (17, 13, 0), (17, 13, 30000), (0, 0, 0)):
            ctx.tune_split(big); ctx.tune_sort(lb)
            for ns in (0, 5):                                   # 0: the automatic slabs, 4-byte records; explicit: 8-byte records
                ctx.tune(wbits=wb, nslabs=ns)
                assert (sppark_amd.to_affine(ctx.invoke(pts, s_, ffi_affine_sz=96)) == exp).all(), (wb, lb, big, ns)
    ctx.close()


@pytest.mark.parametrize("curve,name", CURVES)
def test_msm_full_size_vs_oracle(oracle, libs, curve, name):
    """BASELINE size (2^26 points; configs[2] BLS12-381 G1 and configs[4] alt_bn128 G1) against the
    ORACLE through the period of the inputs: with 2048 distinct points replicated cyclically (the
    shape of poc/msm-cuda/src/util.rs:11-38) MSM_n(P, s) = MSM_2048(B, fold(s)) where fold(s)_j is
    the sum of the scalars of class j mod r (oracle/fold.py).  Checked for
      (i)  periodic scalars with the recipe's edge rows (0, r-1, (r+-1)/2, duplicates, P and -P),
      (ii) INDEPENDENT UNIFORM scalars on [0, r) -- the bench's headline workload --
    and (iii) whole == half + half on the uniform run.
    What these can and cannot see: a gather-index error that is a multiple of the period reads the SAME point -- with
    2048 that is every power-of-two slip from 2^11 on (a wrong index group of the 4-byte sort records, +-k 2^22 here; a
    32-bit wrap of an index or of index * 128).  So (iv): the uniform run again on the same 2^26 scalars with the points
    replicated with the odd prime period 2039 (ragged: 2^26 = 32912 * 2039 + 432), where k 2^m = 0 mod 2039 only for
    k = 0 mod 2039 -- any index slip below 2039 * 2^m lands on a different point."""
    import torch
    import sppark_amd
    from sppark_amd import synth
    from oracle import fold
    O = oracle
    lg, per = 26, 2048
    n = 1 << lg
    r = O.FR_MODULUS[curve]
    base, sc = recipe.msm_inputs(curve, per, 2626, ndistinct=per, edge=True)
    d_base = torch.from_numpy(base).cuda(); d_sc = torch.from_numpy(sc).cuda()
    idx = torch.arange(n, device="cuda") % per
    pts = d_base[idx].contiguous(); scal = d_sc[idx].contiguous()
    # device-resident inputs produced by torch kernels: run on torch's stream so that the MSM is
    # ordered after them
    ctx = sppark_amd.MsmContext(name, stream=torch.cuda.current_stream().cuda_stream)
    got = sppark_amd.to_affine(ctx.invoke(pts, scal), name)
    assert (got == O.msm_affine(curve, base, fold.fold_scalars(scal, per, r), algo=0, param=8)).all()
    del scal
    rnd = synth.uniform_scalars(n, name, seed=3)
    whole = ctx.invoke(pts, rnd)
    assert (sppark_amd.to_affine(whole, name) == O.msm_affine(curve, base, fold.fold_scalars(rnd, per, r), algo=0, param=8)).all()
    h = n // 2
    parts = np.stack([ctx.invoke(pts[:h], rnd[:h]), ctx.invoke(pts[h:], rnd[h:])])
    assert (sppark_amd.to_affine(sppark_amd.jacobian_sum(parts, name), name) == sppark_amd.to_affine(whole, name)).all()
    del pts
    odd = 2039
    pts = d_base[torch.arange(n, device="cuda") % odd].contiguous()
    got = sppark_amd.to_affine(ctx.invoke(pts, rnd), name)
    assert (got == O.msm_affine(curve, base[:odd], fold.fold_scalars(rnd, odd, r), algo=0, param=8)).all()
    assert not (got == sppark_amd.to_affine(whole, name)).all()
    ctx.close()


@pytest.mark.parametrize("curve,name", [(0, "bls12_381"), (1, "bn254")])
def test_msm_all_distinct_points_full_size(oracle, libs, curve, name):
    """2^26 points that do NOT repeat (the reference's test holds its result against an oracle on arbitrary points,
    poc/msm-cuda/tests/msm.rs:19-39; no CPU oracle finishes 2^26): P_i = (a + i b) G generated and normalised on the
    device (sppark_g1_generate_progression; sampled against the oracle's scalar multiplication here and in
    test_generate_progression_matches_oracle), independent uniform scalars, expected result (sum s_i (a + i b) mod r) G
    from integer arithmetic on the scalars and ONE oracle scalar multiplication.  ANY wrong gather index i' changes the
    sum by s_i b (i' - i) G != 0.  Also all scalars equal (sum = s (n a + b n (n - 1) / 2) G: every window one level-A
    partition) and the first 2^25 + 12345 points (a ragged prefix: other plan, other index groups)."""
    import torch
    import sppark_amd
    from sppark_amd import synth
    from oracle import fold
    O = oracle
    fb = O.FP_BYTES[curve]
    r = O.FR_MODULUS[curve]
    n = 1 << 26
    a, b = 0x243f6a8885a308d313198a2e03707344, 0xa4093822299f31d0082efa99       # < 2^126, < 2^96
    G = O.g1_generator(curve)
    pts = torch.empty((n, 2 * fb), dtype=torch.uint8, device="cuda")
    sppark_amd.generate_progression(pts, n, a, b, 2 * fb, name)
    for i in (0, 12345678, n - 1):
        assert (pts[i].cpu().numpy() == O.g1_mul(curve, G, a + i * b)).all()
    sc = synth.uniform_scalars(n, name, seed=2600 + curve)
    ctx = sppark_amd.MsmContext(name, stream=torch.cuda.current_stream().cuda_stream)

    def check(p_, s_, what):
        s0, s1 = fold.weighted_sums(s_)
        exp = O.g1_mul(curve, G, (a * s0 + b * s1) % r)
        assert (sppark_amd.to_affine(ctx.invoke(p_, s_), name) == exp).all(), (name, what)

    check(pts, sc, "uniform")
    m = (1 << 25) + 12345
    check(pts[:m], sc[:m], "ragged prefix")
    eq = sc.clone(); eq[:] = sc[0]
    check(pts, eq, "all equal")
    ctx.close()


def test_msm_above_2p28_vs_oracle(oracle, libs):
    """BASELINE configs[3]'s TOTAL size on one device: 2^28 + 3 * 2048 points (not a power of two) through the plain
    path in one piece (about 150 GB of inputs and scratch), against the oracle through the period fold.  Every
    32-bit index of the driver and the kernels -- window * n offsets, 12 x 2^28 sort entries, 2^16-entry level-A
    partitions beyond level B's register path -- runs four times above anything the other tests reach."""
    import torch
    import sppark_amd
    from sppark_amd import synth
    from oracle import fold
    O = oracle
    per = 2048
    n = (1 << 28) + 3 * per
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 200 << 30:
        pytest.skip("needs ~150 GB of device memory")
    pts, base = synth.replicated_points(n, "bls12_381", per, 0x5eed5eed0001)
    sc = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    for k, lo in enumerate(range(0, n, 1 << 25)):                # pieces: bounded temporaries of the generator
        m = min(1 << 25, n - lo)
        sc[lo:lo + m] = synth.uniform_scalars(m, "bls12_381", 280 + k)
    ctx = sppark_amd.MsmContext("bls12_381", stream=torch.cuda.current_stream().cuda_stream)
    out = ctx.invoke(pts, sc)
    assert ctx.last_chunks() == 1
    exp = O.msm_affine(O.BLS12_381, base.cpu().numpy(), fold.fold_scalars(sc, per, O.FR_MODULUS[O.BLS12_381]), algo=0, param=8)
    assert (sppark_amd.to_affine(out) == exp).all()
    del pts
    # the same scalars over points of the odd prime period 2039 (see test_msm_full_size_vs_oracle: 2048 cannot see an index
    # slip of k 2^m, m >= 11 -- at this size that includes index bit 28, the bit no other test sets)
    odd = 2039
    pts = base[torch.arange(n, device="cuda") % odd].contiguous()
    out2 = ctx.invoke(pts, sc)
    exp2 = O.msm_affine(O.BLS12_381, base[:odd].cpu().numpy(), fold.fold_scalars(sc, odd, O.FR_MODULUS[O.BLS12_381]), algo=0, param=8)
    assert (sppark_amd.to_affine(out2) == exp2).all() and not (exp2 == exp).all()
    ctx.close()
    del pts, sc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("per", [2048, 2039])
@pytest.mark.parametrize("lg", [23, 26])
def test_msm_skewed_scalars_full_size_vs_oracle(oracle, libs, lg, per):
    """Skewed scalars at the sizes where the 4-byte sort records need their upper index bits back (2^23: two index groups,
    2^26: sixteen, four slabs each): ALL scalars equal -- every window is ONE level-A partition of n entries, sorted by the
    cooperative kernels in 64 slices that cross the group boundaries --, every second scalar zero (the groups' first
    positions are no longer multiples of anything), 16-bit scalars (every window but one empty) and a mix; not a power of
    two at 2^23.  Against the oracle through the period fold (oracle/fold.py).
    What each period detects: with 2048 distinct points (the reference's shape) the test sees a lost or duplicated ENTRY,
    a wrong sign, a wrong bucket -- but NOT a wrong index group, which moves an index by a multiple of 2^IB >= 2^18 and
    gathers the same point.  The period 2039 (odd prime; the sizes are then ragged) is the leg that fails on a wrong
    group, on any k 2^m index slip, on a 32-bit wrap of index * stride."""
    import torch
    import sppark_amd
    from sppark_amd import synth
    from oracle import fold
    O = oracle
    n = (1 << lg) + (5 * per if lg == 23 else 0)
    r = O.FR_MODULUS[O.BLS12_381]
    pts, base = synth.replicated_points(n, "bls12_381", per, 0x5eed5eed0002)
    sc = synth.uniform_scalars(n, "bls12_381", 230 + lg)
    ctx = sppark_amd.MsmContext("bls12_381", stream=torch.cuda.current_stream().cuda_stream)
    base_np = base.cpu().numpy()

    def check(s_, what):
        got = sppark_amd.to_affine(ctx.invoke(pts, s_))
        assert (got == O.msm_affine(O.BLS12_381, base_np, fold.fold_scalars(s_, per, r), algo=0, param=8)).all(), (lg, what)

    eq = sc.clone(); eq[:] = sc[0]
    check(eq, "all equal")
    if lg == 23:
        half = sc.clone(); half[::2] = 0
        check(half, "every second scalar zero")
        s16 = torch.zeros_like(sc); s16[:, :2] = sc[:, :2]
        check(s16, "16-bit scalars")
        mix = sc.clone(); mix[n // 16: n - n // 8] = sc[1]
        check(mix, "13/16 equal, uniform at both ends")
    ctx.close()
    del pts, sc, eq
    torch.cuda.empty_cache()


@pytest.mark.parametrize("curve,name,lg,per", [(0, "bls12_381", 23, 2048), (1, "bn254", 24, 2048), (0, "bls12_381", 26, 2048),
                                                (0, "bls12_381", 23, 2039), (0, "bls12_381", 26, 2039)])
def test_msm_fixed_base_full_size_vs_oracle(oracle, libs, curve, name, lg, per):
    """the fixed-base mode at the sizes it builds its tables by itself (>= 2^23 points: automatic window, 2^12
    staged level-A partitions, the cooperative level B in LDS-staged slices on ALL partitions), against the ORACLE
    through the period of the inputs: periodic scalars with the recipe's edge rows, then independent uniform scalars;
    the size is NOT a power of two (ragged last slab / slice / run), and a prefix falls back to the plain path.
    per = 2039 (odd prime): the legs that can see an index slip of k 2^m in the (digit, multiple) pairs and the table
    gathers (2048 gathers the same point for every m >= 11)."""
    import torch
    import sppark_amd
    from sppark_amd import synth
    from oracle import fold
    O = oracle
    n = (1 << lg) + 3 * 2048
    r = O.FR_MODULUS[curve]
    base, sc = recipe.msm_inputs(curve, per, 2323 + lg, ndistinct=per, edge=True)
    d_base = torch.from_numpy(base).cuda(); d_sc = torch.from_numpy(sc).cuda()
    idx = torch.arange(n, device="cuda") % per
    pts = d_base[idx].contiguous(); scal = d_sc[idx].contiguous()
    ctx = sppark_amd.MsmContext(name, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_points(pts, fixed_base=True)
    del pts
    assert ctx.fixed_base_windows() >= 10 and ctx.preloaded() == n
    got = sppark_amd.to_affine(ctx.invoke(None, scal), name)
    assert (got == O.msm_affine(curve, base, fold.fold_scalars(scal, per, r), algo=0, param=8)).all()
    del scal
    rnd = synth.uniform_scalars(n, name, seed=5)
    whole = sppark_amd.to_affine(ctx.invoke(None, rnd), name)
    assert (whole == O.msm_affine(curve, base, fold.fold_scalars(rnd, per, r), algo=0, param=8)).all()
    m = (n // 2 // 2048) * 2048                                  # a prefix: the plain path on the tables' first level
    assert (sppark_amd.to_affine(ctx.invoke(None, rnd[:m].contiguous(), npoints=m), name)
            == O.msm_affine(curve, base, fold.fold_scalars(rnd[:m].contiguous(), per, r), algo=0, param=8)).all()
    ctx.close()


# ------------------------------------------------- bucket sums: chunked levels vs the subset-sum top
@pytest.mark.parametrize("curve,name", CURVES)
def test_msm_bucket_sum_top(oracle, libs, curve, name):
    """k_bucket_top_bits / k_bucket_top_sum against the chunked levels they replace (the G2 entry points
    run them with the automatic hand-over; test_msm_g2_*): every
    hand-over point (never, 32 .. 32768 partial sums per window), window sizes that give short and long
    windows, bucket chunk 4 and 8, and a skewed case where most partial sums are the point at infinity."""
    import sppark_amd
    O = oracle
    n = 30000
    pts, sc = recipe.msm_inputs(curve, n, 4242, ndistinct=700, flagged=True)
    sc_skew = sc.copy(); sc_skew[: n - 50, 2:] = 0              # 16-bit scalars: the upper windows are empty
    cases = [(name, curve, pts, sc), (name, curve, pts, sc_skew)]
    for ctx_name, oid, PT, SC in cases:
        exp = O.msm_affine(oid, PT, SC, algo=0, param=8)
        ctx = sppark_amd.MsmContext(ctx_name)
        for wb in (0, 11, 16, 18):
            for K in (0, 4, 8):
                ctx.tune(wbits=wb, K=K)
                for top in (1, 32, 512, 0, 32768):
                    ctx.tune_sums(top)
                    out = ctx.invoke(PT, SC, ffi_affine_sz=PT.shape[1])
                    assert (sppark_amd.to_affine(out, ctx_name) == exp).all(), (ctx_name, wb, K, top)
        # the cooperative top runs a work-group per PIECE of a sum (bucket_top_piece: the plain sum of a top of 4096 items and
        # more in two pieces -- wb 16 / 18 above); tail variant 7: per sum
        ctx.tune_tail(7, 0)
        for wb, top in ((16, 0), (18, 32768), (13, 512)):
            ctx.tune(wbits=wb); ctx.tune_sums(top)
            out = ctx.invoke(PT, SC, ffi_affine_sz=PT.shape[1])
            assert (sppark_amd.to_affine(out, ctx_name) == exp).all(), (ctx_name, wb, top, "per sum")
        ctx.close()


# ------------------------------------------------- pipeline shape: window groups, chunks, devices
@pytest.mark.parametrize("curve,name", CURVES)
def test_msm_window_groups_and_chunks(oracle, libs, curve, name):
    """The two-stream window-group pipeline and the chunked path, forced through every shape on
    sizes the oracle finishes quickly: groups 1..all windows (uneven last group included), chunks
    that do not divide n, host and device inputs, flagged and plain points, preloaded bases, and a
    scratch bound that forces the bounded-memory path."""
    import torch
    import sppark_amd
    O = oracle
    n = 20000
    pts, sc = recipe.msm_inputs(curve, n, 777, ndistinct=500, flagged=True)
    exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
    plain = np.ascontiguousarray(pts[:, :-8])                   # Affine_t: infinity = all-zero coordinates
    plain[(pts[:, -8] != 0)] = 0
    d_pts = torch.from_numpy(pts).cuda(); d_sc = torch.from_numpy(sc).cuda(); d_plain = torch.from_numpy(plain).cuda()
    ctx = sppark_amd.MsmContext(name)
    for wb in (0, 9, 13):
        ctx.tune(wbits=wb)
        nw = ctx.plan(n)["windows"]
        for groups in sorted({1, 2, 3, 5, nw}):
            for chunk in (0, 7001, 4096):
                ctx.tune_pipeline(groups=groups, chunk_points=chunk)
                for p_, s_, st in ((pts, sc, pts.shape[1]), (d_pts, d_sc, pts.shape[1]), (d_plain, sc, plain.shape[1]), (plain, d_sc, plain.shape[1])):
                    out = ctx.invoke(p_, s_, ffi_affine_sz=st)
                    assert (sppark_amd.to_affine(out, name) == exp).all(), (name, wb, groups, chunk, st)
                    assert ctx.last_chunks() == (1 if chunk == 0 else -(-n // chunk))
    # preloaded bases through the chunked path
    ctx.tune(wbits=0); ctx.tune_pipeline(groups=3, chunk_points=6000)
    ctx.set_points(plain, ffi_affine_sz=plain.shape[1])
    assert (sppark_amd.to_affine(ctx.invoke(None, sc), name) == exp).all()
    assert (sppark_amd.to_affine(ctx.invoke(None, d_sc[:5000], npoints=5000), name) == O.msm_affine(curve, pts[:5000], sc[:5000], algo=0, param=8)).all()
    ctx.set_points(None)
    # bounded scratch: the chunk is halved until the scratch fits
    ctx.tune_pipeline(groups=0, chunk_points=0, max_scratch_bytes=0)
    ctx.invoke(d_pts, d_sc, ffi_affine_sz=pts.shape[1])
    full = ctx.scratch_bytes()
    ctx2 = sppark_amd.MsmContext(name)
    ctx2.tune_pipeline(max_scratch_bytes=full // 3)
    out = ctx2.invoke(d_pts, d_sc, ffi_affine_sz=pts.shape[1])
    assert (sppark_amd.to_affine(out, name) == exp).all()
    assert ctx2.last_chunks() > 1 and ctx2.scratch_bytes() <= full // 3
    ctx.close(); ctx2.close()


@pytest.mark.parametrize("per", [512, 509])
def test_msm_pipeline_medium_size(oracle, libs, per):
    """2^22 points (4 window groups on two streams forced; host inputs: 4 chunks) against
    the oracle through the period fold, device- and host-resident, repeated back to back so that a
    missing event between the streams would show.  Period 512 and the odd prime 509 (a chunk offset or an
    index slip of k 2^m is a different point only under the latter)."""
    import torch
    import sppark_amd
    from sppark_amd import synth
    from oracle import fold
    O = oracle
    n = 1 << 22
    base, _ = recipe.msm_inputs(O.BLS12_381, per, 99, ndistinct=per, edge=True)
    d_base = torch.from_numpy(base).cuda()
    pts = d_base[torch.arange(n, device="cuda") % per].contiguous()
    ctx = sppark_amd.MsmContext("bls12_381", stream=torch.cuda.current_stream().cuda_stream)
    assert ctx.plan_groups(n) == 1
    ctx.tune_pipeline(groups=4)
    assert ctx.plan_groups(n) == 4
    for it in range(3):
        sc = synth.uniform_scalars(n, "bls12_381", seed=10 + it)
        exp = O.msm_affine(O.BLS12_381, base, fold.fold_scalars(sc, per, O.FR_MODULUS[O.BLS12_381]), algo=0, param=8)
        assert (sppark_amd.to_affine(ctx.invoke(pts, sc)) == exp).all(), it
        if it == 0:
            h_pts, h_sc = pts.cpu().numpy(), sc.cpu().numpy()
            assert (sppark_amd.to_affine(ctx.invoke(h_pts, h_sc)) == exp).all()
            assert ctx.last_chunks() == 4                        # 2^22 host points: 2^20 per chunk
            assert (sppark_amd.to_affine(sppark_amd.multi_scalar_mult(h_pts, h_sc)) == exp).all()
    ctx.close()


def test_msm_multi_device_entry_points(oracle, libs):
    """sppark_msm_multi / sppark_msm_multi_shards: one host thread and one pooled context per shard.
    On a single-GPU box the shards share device 0 (a device may be named more than once), which
    still exercises the threads, the context pool and the host combine; uneven and empty shards and
    a shard whose partial sum is infinity included."""
    import torch
    import sppark_amd
    O = oracle
    nd = sppark_amd.ngpus()
    assert nd >= 1
    n = 9001
    pts, sc = recipe.msm_inputs(O.BLS12_381, n, 31337, ndistinct=200)
    exp = O.msm_affine(O.BLS12_381, pts, sc, algo=0, param=8)
    for ndev in sorted({0, 1, nd}):
        assert (sppark_amd.to_affine(sppark_amd.msm_multi(pts, sc, ndev=ndev)) == exp).all()
    with pytest.raises(sppark_amd.SpparkError):
        sppark_amd.msm_multi(pts, sc, ndev=nd + 1)
    cuts = [0, 1000, 1000, 4567, n]                              # an empty shard in the middle
    zsc = sc.copy(); zsc[1000:4567] = 0                          # ... and one that sums to infinity
    for s_, e_ in ((sc, exp), (zsc, O.msm_affine(O.BLS12_381, pts, zsc, algo=0, param=8))):
        shards = [(pts[a:b], s_[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        ids = [i % nd for i in range(len(shards))]
        assert (sppark_amd.to_affine(sppark_amd.msm_multi_shards(shards, device_ids=ids)) == e_).all()
        dev = [(torch.from_numpy(p).to("cuda:%d" % i), torch.from_numpy(s).to("cuda:%d" % i)) for (p, s), i in zip(shards, ids)]
        torch.cuda.synchronize()
        assert (sppark_amd.to_affine(sppark_amd.msm_multi_shards(dev, device_ids=ids)) == e_).all()
        # the timed variants: same result, one wall-clock figure per shard (0 for the empty one)
        out, ms = sppark_amd.msm_multi_shards(shards, device_ids=ids, timings=True)
        assert (sppark_amd.to_affine(out) == e_).all() and len(ms) == len(shards)
        assert ms[1] == 0.0 and all(v > 0 for i, v in enumerate(ms) if i != 1)
    out, ms = sppark_amd.msm_multi(pts, sc, ndev=1, timings=True)
    assert (sppark_amd.to_affine(out) == exp).all() and len(ms) == 1 and ms[0] > 0


def test_msm_rccl_exchange_single_rank(oracle, libs):
    """sppark_msm_rccl / sppark_msm_rccl_sum: the native one-process-per-GPU exchange over an RCCL communicator made
    without torch (ncclGetUniqueId + ncclCommInitRank through ctypes, as a Rust / Go caller would).  One GPU in the pool:
    a communicator of ONE rank -- the all-gather, the staging and the host sum run, the collective has no peer.
    G1 from host and device buffers, an empty shard, a G2 partial sum, the error path with the communicator intact."""
    import torch
    import sppark_amd
    from sppark_amd import multi_gpu
    O = oracle
    comm = multi_gpu.RcclComm(1, 0, multi_gpu.RcclComm.unique_id())
    try:
        n = 7001
        pts, sc = recipe.msm_inputs(O.BLS12_381, n, 2718, ndistinct=300, flagged=True)
        exp = O.msm_affine(O.BLS12_381, pts, sc, algo=0, param=8)
        assert (sppark_amd.to_affine(multi_gpu.msm_rccl(pts, sc, comm, ffi_affine_sz=pts.shape[1])) == exp).all()
        d_pts, d_sc = torch.from_numpy(pts).cuda(), torch.from_numpy(sc).cuda()
        torch.cuda.synchronize()
        assert (sppark_amd.to_affine(multi_gpu.msm_rccl(d_pts, d_sc, comm, ffi_affine_sz=pts.shape[1])) == exp).all()
        assert (multi_gpu.msm_rccl(pts[:0], sc[:0], comm, ffi_affine_sz=pts.shape[1]) == 0).all()      # a rank without points
        part = sppark_amd.multi_scalar_mult_arkworks(pts, sc)
        assert (multi_gpu.rccl_sum(part, comm) == part).all()
        # IN PLACE (out == partial, what a caller does right after an MSM entry point filled `out`): the partial sum is
        # read before `out` is cleared
        from sppark_amd import ffi
        inplace = part.copy()
        ffi.check(ffi.load("bls12_381"), ffi.load("bls12_381").sppark_msm_rccl_sum(inplace.ctypes.data, inplace.ctypes.data, 0, comm.handle, None))
        assert (sppark_amd.to_affine(inplace) == exp).all()
        name2, curve2 = G2[0]
        p2, s2 = recipe.msm_inputs(curve2, 65, 99, ndistinct=16, flagged=True)
        part2 = sppark_amd.multi_scalar_mult_fp2_arkworks(p2, s2, name2)
        assert (sppark_amd.to_affine_g2(multi_gpu.rccl_sum(part2, comm, name2, g2=True), name2) == sppark_amd.to_affine_g2(part2, name2)).all()
        # a local failure (a stride below two field elements) is reported with `out` at infinity; the collective still ran
        L = ffi.load("bls12_381")
        out = np.full(144, 0xff, dtype=np.uint8)
        err = L.sppark_msm_rccl(out.ctypes.data, pts.ctypes.data, n, sc.ctypes.data, 0, 8, comm.handle, None)
        assert err.code != 0 and (out == 0).all()
        L.drop_error_message(err.message)
        assert (sppark_amd.to_affine(multi_gpu.msm_rccl(pts, sc, comm, ffi_affine_sz=pts.shape[1])) == exp).all()
    finally:
        comm.destroy()


def test_msm_rccl_exchange_many_ranks_with_a_test_double(oracle, libs, tmp_path):
    """The N > 1 shape of sppark_msm_rccl on one GPU: RCCL refuses two ranks on one device, so the three entry points
    the library binds are supplied by a TEST DOUBLE (tests/emu/fake_rccl.cpp, via SPPARK_RCCL_LIB) whose ranks are
    threads of one process and whose all-gather is N device-to-device copies behind a barrier.  Four ranks with uneven
    shards, an empty one included: every rank returns the whole MSM; then rank 2 fails locally (bad stride): it reports
    its error with `out` at infinity, the others do not hang and return the sum of the remaining shards."""
    import json
    import subprocess
    import sys
    O = oracle
    here = os.path.dirname(os.path.abspath(__file__))
    lib = str(tmp_path / "libfake_rccl.so")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib,
                    os.path.join(here, "emu", "fake_rccl.cpp")], check=True, capture_output=True)
    n = 7001
    pts, sc = recipe.msm_inputs(O.BLS12_381, n, 161803, ndistinct=300, flagged=True)
    np.savez(tmp_path / "in.npz", pts=pts, sc=sc)
    cuts = [0, 1000, 1000, 4567, n]
    env = dict(os.environ, SPPARK_RCCL_LIB=lib)
    for bad in (-1, 2):
        r = subprocess.run([sys.executable, os.path.join(here, "rccl_ranks_worker.py"), str(tmp_path / "in.npz"), json.dumps(cuts), str(bad)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
        res = json.loads(r.stdout.strip().splitlines()[-1])
        assert not any(res["hung"])
        keep = np.ones(n, dtype=bool)
        if bad >= 0:
            keep[cuts[bad]:cuts[bad + 1]] = False
        zsc = sc.copy(); zsc[~keep] = 0
        exp = O.msm_affine(O.BLS12_381, pts, zsc, algo=0, param=8)
        for rank, (code, out) in enumerate(zip(res["codes"], res["outs"])):
            out = np.frombuffer(bytes.fromhex(out), dtype=np.uint8)
            if rank == bad:
                assert code != 0 and (out == 0).all()
            else:
                import sppark_amd
                assert code == 0 and (sppark_amd.to_affine(out) == exp).all(), (bad, rank)


def test_one_shot_pool_is_not_tied_to_threads(oracle, libs):
    """The one-shot entry points borrow contexts from a process-wide pool: calls from short-lived
    threads reuse them (no scratch stranded per dead thread), concurrent calls get distinct ones."""
    import threading
    import sppark_amd
    from sppark_amd import ffi
    O = oracle
    L = ffi.load("bls12_381")
    pts, sc = recipe.msm_inputs(O.BLS12_381, 5000, 4711, flagged=True)
    exp = O.msm_affine(O.BLS12_381, pts, sc, algo=0, param=8)
    bad = []

    def work():
        for _ in range(3):
            if not (sppark_amd.to_affine(sppark_amd.multi_scalar_mult_arkworks(pts, sc)) == exp).all():
                bad.append(1)
    for rounds in range(3):                                     # generations of threads that exit
        ts = [threading.Thread(target=work) for _ in range(4)]
        for t in ts: t.start()
        for t in ts: t.join()
    assert not bad
    L.sppark_msm_release_cached()
    assert (sppark_amd.to_affine(sppark_amd.multi_scalar_mult_arkworks(pts, sc)) == exp).all()


@pytest.mark.parametrize("curve,name", CURVES)
def test_batch_addition_bitmaps(oracle, libs, curve, name):
    """sppark_batch_addition (msm/batch_addition.cuh:25-132): sum over a bitmap, and the difference
    of two selections with a reference map.  Expectation = the oracle's MSM with scalars 1 / r-1 / 0
    and, for the small case, the independent Python group law; densities from empty to full,
    ragged tail, infinity inputs, duplicated points (doubling branch), host and device buffers."""
    import torch
    import pygroup
    import sppark_amd
    O = oracle
    r = O.FR_MODULUS[curve]
    one = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
    neg = np.frombuffer((r - 1).to_bytes(32, "little"), dtype=np.uint8)
    rng = np.random.default_rng(77 + curve)
    for n, nd, flagged in ((1, 1, False), (31, 31, True), (32, 8, False), (33, 33, True), (1000, 3, False), (70001, 500, True)):
        pts, _ = recipe.msm_inputs(curve, n, 900 + n, ndistinct=nd, flagged=flagged)
        words = (n + 31) // 32
        for density in (0.0, 0.03, 0.5, 1.0):
            bm = (rng.random(words * 32) < density)
            rm = (rng.random(words * 32) < 0.3)
            for use_ref in (False, True):
                sel = bm ^ rm if use_ref else bm
                sc = np.zeros((n, 32), dtype=np.uint8)
                sc[sel[:n]] = one
                if use_ref:
                    sc[(sel & rm)[:n]] = neg
                exp = O.msm_affine(curve, pts, sc, algo=0, param=4)
                bw = np.packbits(bm.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
                rw = np.packbits(rm.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
                assert all(((int(bw[i // 32]) >> (i % 32)) & 1) == int(bm[i]) for i in (0, min(n - 1, 37)))
                out = sppark_amd.batch_addition(pts, bw, rw if use_ref else None, name, ffi_affine_sz=pts.shape[1])
                assert (sppark_amd.to_affine(out, name) == exp).all(), (name, n, density, use_ref)
                if n == 33:
                    pin = pygroup.msm_affine_bytes(name, False, pts.tobytes(), pts.shape[1], flagged, sc.tobytes())
                    assert pin == exp.tobytes()
                if n == 70001 and density == 0.5:
                    d = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
                    out = sppark_amd.batch_addition(d(pts), d(bw), d(rw) if use_ref else None, name, ffi_affine_sz=pts.shape[1])
                    assert (sppark_amd.to_affine(out, name) == exp).all()
