"""Polynomial primitives: host API over sppark_prefix_op / sppark_poly_evaluate /
sppark_div_by_x_minus_z (the reference offers them as C++ templates only:
polynomial/prefix_op.cuh:324-396, evaluate.cuh:307-412, div_by_x_minus_z.cuh:447-486).
Buffers are numpy arrays or torch tensors (host or device) of field elements in the wire
format of the chosen library (gl64: canonical u64; bb31: Montgomery u32; bls12_381 / bn254:
the curve's scalar field, 4 x u64 Montgomery)."""
from . import ffi

_ELEM_BYTES = {"gl64": 8, "bb31": 4, "bls12_381": 32, "bn254": 32, "bls12_377": 32, "pallas": 32, "vesta": 32, "m31": 4, "bb31x4": 16, "gl64_plonky2": 8, "bb31_canonical": 4}
ADD, MULTIPLY = 0, 1


def _count(buf, field):
    nbytes = int(buf.nbytes) if hasattr(buf, "nbytes") else int(buf.numel() * buf.element_size())
    if nbytes % _ELEM_BYTES[field]:
        raise ValueError("buffer is not a whole number of field elements")
    return nbytes // _ELEM_BYTES[field]


def prefix_op(out, inp, op, field="gl64", device_id=0, stream=None):
    """out[i] = inp[0] (op) ... (op) inp[i]; op = ADD or MULTIPLY; out may be inp itself."""
    L = ffi.load(field)
    n = _count(inp, field)
    if _count(out, field) != n:
        raise ValueError("length mismatch")
    po, _k1 = ffi.as_pointer(out); pi, _k2 = ffi.as_pointer(inp)
    ffi.check(L, L.sppark_prefix_op(device_id, po, pi, n, int(op), stream))
    return out


def evaluate(ret, xs, coeffs, field="gl64", device_id=0, stream=None):
    """ret[j] = sum_i coeffs[i] * xs[j]^i"""
    L = ffi.load(field)
    n = _count(xs, field)
    if _count(ret, field) != n:
        raise ValueError("length mismatch")
    pr, _k1 = ffi.as_pointer(ret); px, _k2 = ffi.as_pointer(xs); pc, _k3 = ffi.as_pointer(coeffs)
    ffi.check(L, L.sppark_poly_evaluate(device_id, pr, px, n, pc, _count(coeffs, field), stream))
    return ret


def div_by_x_minus_z(inout, z, rotate=False, field="gl64", device_id=0, stream=None):
    """In-place synthetic division of sum_i inout[i] x^i by (x - z).  rotate=False: remainder at
    index 0, quotient after it; rotate=True: quotient first, remainder last."""
    L = ffi.load(field)
    if _count(z, field) != 1:
        raise ValueError("z must be one field element")
    p, _k1 = ffi.as_pointer(inout); pz, _k2 = ffi.as_pointer(z)
    ffi.check(L, L.sppark_div_by_x_minus_z(device_id, p, _count(inout, field), pz, int(rotate), stream))
    return inout
