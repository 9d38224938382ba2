"""GPU parity tests for the polynomial primitives (sppark_prefix_op, sppark_poly_evaluate,
sppark_div_by_x_minus_z: the reference's polynomial/*.cuh), through the C ABI.  Field elements have
unique bit patterns, so equality is exact."""
import json
import os

import numpy as np
import pytest

import recipe

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WIDE = ("bls12_381", "bn254", "bls12_377", "pallas", "vesta")
FIELDS = ["gl64", "bb31"] + list(WIDE)


def _arr(c, key):
    f = c["field"]
    dt = np.uint32 if f in ("bb31", "m31", "bb31x4") else np.uint64
    a = np.frombuffer(bytes.fromhex(c[key]), dtype=dt).copy()
    return a.reshape(-1, 4) if (f in WIDE or f == "bb31x4") else a


def test_poly_golden_vectors(libs):
    """definition-level Python big-int vectors (tests/golden/make_golden.py::make_poly)"""
    from sppark_amd import poly
    for c in json.load(open(os.path.join(HERE, "golden", "poly_golden.json"))):
        f = c["field"]
        coeffs, z = _arr(c, "coeffs"), _arr(c, "z")
        for rot, key in ((False, "div"), (True, "div_rotate")):
            x = coeffs.copy()
            poly.div_by_x_minus_z(x, z, rotate=rot, field=f)
            assert (x == _arr(c, key)).all(), (f, c["len"], key)
        if "prefix_add" in c:
            for op, key in ((poly.ADD, "prefix_add"), (poly.MULTIPLY, "prefix_mul")):
                out = np.zeros_like(coeffs)
                poly.prefix_op(out, coeffs, op, field=f)
                assert (out == _arr(c, key)).all(), (f, c["len"], key)
                x = coeffs.copy()
                poly.prefix_op(x, x, op, field=f)                       # in place
                assert (x == _arr(c, key)).all(), (f, c["len"], key, "in place")
            xs = _arr(c, "xs")
            ret = np.zeros_like(xs)
            poly.evaluate(ret, xs, coeffs, field=f)
            assert (ret == _arr(c, "evaluate")).all(), (f, c["len"])


def _pool(field, lg):
    """2^lg random field elements in the wire format of |field|"""
    if field == "m31":
        return (np.random.default_rng(3).integers(0, (1 << 31) - 1, size=1 << lg, dtype=np.uint64)).astype(np.uint32)
    if field == "bb31x4":
        return recipe.ntt_input("bb31", lg + 2, 98).reshape(-1, 4)
    if field in WIDE:                                           # vectorised: values < 2^252 < r
        rng = np.random.default_rng(5)
        pool = rng.integers(0, 1 << 63, size=(1 << lg, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(1 << lg, 4), dtype=np.uint64)
        pool[:, 3] &= np.uint64(0x0fffffffffffffff)
        return pool
    return recipe.ntt_input(field, lg, 99)


@pytest.mark.parametrize("field", FIELDS + ["m31", "bb31x4"])
def test_poly_vs_oracle(oracle, libs, field):
    """lengths around lane / tile / spine edges and a large one, host and device buffers"""
    import torch
    from sppark_amd import poly
    O = oracle
    wide = field in WIDE or field == "bb31x4"
    tile = 1024 if wide else 2048
    lens = [1, 3, 255, 256, 257, tile - 1, tile, tile + 1, 3 * tile + 5, 256 * tile, 256 * tile + 1, 257 * tile + 77]
    lens.append((1 << 20) + 3 if wide else (1 << 22) + 3)
    lg = max(lens).bit_length()
    pool = _pool(field, lg)
    z = pool[7:8].copy()
    xs = pool[100:106].copy()
    for n in lens:
        c = pool[:n].copy()
        for rot in (False, True):
            x = c.copy()
            poly.div_by_x_minus_z(x, z, rotate=rot, field=field)
            assert (x == O.div_by_x_minus_z(field, c, z, rotate=rot)).all(), (field, n, rot)
        for op in (poly.ADD, poly.MULTIPLY):
            out = np.zeros_like(c)
            poly.prefix_op(out, c, op, field=field)
            assert (out == O.prefix_op(field, c, op)).all(), (field, n, op)
        ret = np.zeros_like(xs)
        poly.evaluate(ret, xs, c, field=field)
        assert (ret == O.poly_evaluate(field, c, xs)).all(), (field, n)
    # device-resident buffers on torch's stream
    n = lens[-1]
    c = pool[:n].copy()
    view = np.int32 if field in ("bb31", "m31", "bb31x4") else np.int64
    d = torch.from_numpy(c.view(view)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    poly.prefix_op(d, d, poly.ADD, field=field, stream=s)
    torch.cuda.synchronize()
    assert (d.cpu().numpy().view(c.dtype).reshape(c.shape) == O.prefix_op(field, c, 0)).all()
    d = torch.from_numpy(c.view(view)).cuda()
    poly.div_by_x_minus_z(d, z, rotate=True, field=field, stream=s)
    torch.cuda.synchronize()
    assert (d.cpu().numpy().view(c.dtype).reshape(c.shape) == O.div_by_x_minus_z(field, c, z, rotate=True)).all()


def test_poly_properties_full_size(libs):
    """Goldilocks, 2^24 coefficients (the NTT's BASELINE size): q(X)*(X - z) + r == p(X) checked at
    random points, remainder == p(z), prefix sums difference == input."""
    from sppark_amd import poly
    P = 0xffffffff00000001
    n = 1 << 24
    c = recipe.ntt_input("gl64", 24, 5)
    z = np.array([0x123456789abcdef], dtype=np.uint64)
    xs = np.array([3, 0xdeadbeef, P - 2], dtype=np.uint64)
    q = c.copy()
    poly.div_by_x_minus_z(q, z, rotate=False, field="gl64")
    pz = np.zeros(1, dtype=np.uint64); poly.evaluate(pz, z, c, field="gl64")
    assert q[0] == pz[0]
    px = np.zeros(3, dtype=np.uint64); poly.evaluate(px, xs, c, field="gl64")
    qx = np.zeros(3, dtype=np.uint64); poly.evaluate(qx, xs, q[1:].copy(), field="gl64")
    for j in range(3):
        assert (int(qx[j]) * (int(xs[j]) - int(z[0])) + int(q[0])) % P == int(px[j])
    s = np.zeros_like(c)
    poly.prefix_op(s, c, poly.ADD, field="gl64")
    d = s[1:] - s[:-1]                                          # mod 2^64
    d = np.where(s[1:] < s[:-1], d + np.uint64(P), d)           # + p where the true difference is negative
    assert s[0] == c[0] and (d == c[1:]).all()
