"""The product's DEVICE FIELD arithmetic against the reference's own device field classes, on the same MI355X.

Under __HIPCC__ the reference defines fp_t / fr_t over ff/mont_t.hip (ff/bls12-381.hpp:63-83, ff/alt_bn128.hpp:60-82,
ff/bls12-377.hpp:61-85, ff/pasta.hpp:55-79); oracle/ref_field_shim.cu compiles exactly those classes from where they lie
(oracle/Makefile: ref_field -> oracle/_ref/libref_field_<curve>.so) and applies their own operators + - * sqr() to()
from() (ff/mont_t.hip:96-218) to arrays.  This is the one layer of the MSM side the reference itself can pin here (its
host field is blst, its point classes and Pippenger are not part of its HIP path): SURVEY section 8 row a1.

Held against it, bit for bit on the memory image (little-endian 32-bit words of the Montgomery form x * 2^(32 n)):
  * ff/mont_dev.hpp   -- the wire-format class (sppark_devtest_field_op), base AND scalar field of every curve;
  * ff/montx_dev.hpp  -- the loosely-reduced 28- / 29-bit-limb class every G1 bucket pipeline computes in, through its own
                         conversions: to_std(op(from_std(a), from_std(b))) must be the reference's a op b.
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CURVES = [(0, "bls12_381"), (1, "bn254"), (4, "bls12_377"), (6, "pallas"), (7, "vesta")]


def P(a):
    return a.ctypes.data


def _need(oracle, name):
    if not oracle.ref_field_available(name):
        # (as tests/test_ntt_vs_reference_gpu.py: the library is built where /root/reference exists and travels prebuilt)
        pytest.skip("oracle/_ref/libref_field_%s.so is not built (oracle/Makefile ref_field needs /root/reference at BUILD time)" % name)


def _vectors(p, nb, n, seed):
    """edge-heavy canonical values: 0, 1, 2, p-1, p-2, R mod p, R^2 mod p, (p-1)/2, powers of two, all-ones masks, random"""
    rng = random.Random(seed)
    R = 1 << (8 * nb)
    edge = [0, 1, 2, p - 1, p - 2, R % p, R * R % p, (p - 1) // 2, (p + 1) // 2, (R - 1) % p, (1 << (8 * nb - 1)) % p]
    edge += [(1 << k) % p for k in (28, 31, 32, 33, 56, 63, 64, 8 * nb - 2)]
    edge += [((1 << k) - 1) % p for k in (28, 32, 64, 8 * nb - 3)]
    vals = [rng.choice(edge) if rng.random() < 0.25 else rng.randrange(p) for _ in range(n)]
    vals[:len(edge)] = edge
    return vals


def _pack(vals, nb):
    return np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in vals), dtype=np.uint8).copy()


@pytest.mark.parametrize("curve,name", CURVES)
def test_wire_field_class_equals_the_reference_field(oracle, libs, curve, name):
    """mont_dev<fp>, mont_dev<fr>: + - * sqr to from, every pairing of the edge values among the operands."""
    from sppark_amd import ffi
    O = oracle
    _need(O, name)
    L = ffi.load_devtest(name)
    for which, p, nb in ((0, O.FP_MODULUS[curve], O.FP_BYTES[curve]), (1, O.FR_MODULUS[curve], 32)):
        n = 4096
        va = _vectors(p, nb, n, 100 * curve + which)
        vb = _vectors(p, nb, n, 100 * curve + which + 50)
        ne = 30                                                 # all pairs of the leading edge values
        pairs = [(va[i], va[j]) for i in range(ne) for j in range(ne)]
        va[64:64 + len(pairs)] = [x for x, _ in pairs]; vb[64:64 + len(pairs)] = [y for _, y in pairs]
        a, b = _pack(va, nb), _pack(vb, nb)
        # (ours, theirs): k_field_op's numbering (csrc/api/devtest_api.hip) vs ref_field_shim.cu's
        for mine, ref, what in ((0, 0, "+"), (1, 1, "-"), (2, 2, "*"), (3, 3, "sqr"), (6, 4, "to"), (5, 5, "from")):
            out = np.zeros_like(a)
            ffi.check(L, L.sppark_devtest_field_op(which, mine, P(out), P(a), P(b), n))
            exp = O.ref_field_op(name, which, ref, a, b)
            bad = np.nonzero((out.reshape(n, nb) != exp.reshape(n, nb)).any(axis=1))[0]
            assert bad.size == 0, (name, "fp" if which == 0 else "fr", what, int(bad[0]), hex(va[bad[0]]), hex(vb[bad[0]]))
        # and the reference itself against big integers on the same vectors (the shim calls what it says it calls)
        R = 1 << (8 * nb)
        exp = O.ref_field_op(name, which, 2, a, b).reshape(n, nb)
        Rinv = pow(R, p - 2, p)
        for i in list(range(40)) + [n - 1]:
            assert int.from_bytes(exp[i].tobytes(), "little") == va[i] * vb[i] * Rinv % p


@pytest.mark.parametrize("curve,name", CURVES)
def test_bucket_field_class_equals_the_reference_field(oracle, libs, curve, name):
    """montx_dev<fp, LB> (14 limbs of 28 bits over the 381 / 377-bit fields, NINE 29-bit limbs over the 254 / 255-bit
    alt_bn128 and Pasta fields): from_std -> op -> to_std
    gives the reference's a op b for * sqr + -, bit for bit; from_std followed by to_std is the identity."""
    from sppark_amd import ffi
    O = oracle
    _need(O, name)
    L = ffi.load_devtest(name)
    NL = L.sppark_devtest_bucket_field_limbs()
    p, nb = O.FP_MODULUS[curve], O.FP_BYTES[curve]
    assert NL == {"bls12_381": 14, "bls12_377": 14, "bn254": 9, "pallas": 9, "vesta": 9}[name]      # 254 / 255 bits: nine 29-bit limbs
    NW = nb // 4
    n = 4096
    va = _vectors(p, nb, n, 7 * curve + 1)
    vb = _vectors(p, nb, n, 7 * curve + 2)
    ne = 30
    pairs = [(va[i], va[j]) for i in range(ne) for j in range(ne)]
    va[64:64 + len(pairs)] = [x for x, _ in pairs]; vb[64:64 + len(pairs)] = [y for _, y in pairs]
    a, b = _pack(va, nb), _pack(vb, nb)

    def widen(x):                                               # wire words in the first NW of NL words per element
        w = np.zeros((n, NL), dtype=np.uint32)
        w[:, :NW] = x.view(np.uint32).reshape(n, NW)
        return w

    def run(op, x, y):
        out = np.zeros((n, NL), dtype=np.uint32)
        ffi.check(L, L.sppark_devtest_bucket_field_op(op, P(out), P(x), P(y), n))
        return out

    def std(x):                                                 # internal -> canonical wire bytes
        return np.ascontiguousarray(run(6, x, x)[:, :NW]).view(np.uint8).reshape(-1)
    ia, ib = run(5, widen(a), widen(a)), run(5, widen(b), widen(b))
    assert (std(ia) == a).all() and (std(ib) == b).all()
    # devtest op (csrc/api/devtest_api.hip: k_bucket_field_op) -> the reference's operator (ref_field_shim.cu)
    for mine, ref, what in ((0, 2, "*"), (1, 3, "sqr"), (3, 0, "+"), (2, 1, "-")):
        got = std(run(mine, ia, ib))
        exp = O.ref_field_op(name, 0, ref, a, b)
        bad = np.nonzero((got.reshape(n, nb) != exp.reshape(n, nb)).any(axis=1))[0]
        assert bad.size == 0, (name, what, int(bad[0]), hex(va[bad[0]]), hex(vb[bad[0]]))
    # a chain that leaves the canonical range on our side and never on theirs: ((a*b - b) + a)^2 * a
    t = run(0, ia, ib); t = run(2, t, ib); t = run(3, t, ia)
    t = run(0, t, t); t = run(0, t, ia)
    r = O.ref_field_op(name, 0, 2, a, b); r = O.ref_field_op(name, 0, 1, r, b); r = O.ref_field_op(name, 0, 0, r, a)
    r = O.ref_field_op(name, 0, 3, r); r = O.ref_field_op(name, 0, 2, r, a)
    assert (std(t) == r).all(), name
