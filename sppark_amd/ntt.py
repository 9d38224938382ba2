"""NTT host API: mirror of poc/ntt-cuda/src/lib.rs:7-118 (NTT, iNTT, coset_NTT,
coset_iNTT) and of the enums in rust/src/lib.rs:99-118 / ntt/ntt.cuh:33-36."""
import enum

from . import ffi


class NTTInputOutputOrder(enum.IntEnum):
    NN = 0
    NR = 1
    RN = 2
    RR = 3


class NTTDirection(enum.IntEnum):
    Forward = 0
    Inverse = 1


class NTTType(enum.IntEnum):
    Standard = 0
    Coset = 1


# one library per field (poc/ntt-cuda/build.rs features): gl64, bb31, and the scalar
# fields of the two curves (256-bit Montgomery elements)
_ELEM_BYTES = {"gl64": 8, "bb31": 4, "bls12_381": 32, "bn254": 32}


def compute_ntt(device_id, inout, order, direction, ntt_type, field="gl64", stream=None):
    """compute_ntt (poc/ntt-cuda/cuda/ntt_api.cu:25-36), in place.  |inout| is a
    numpy array or torch tensor (host or device) of 2^k field elements."""
    L = ffi.load(field)
    nbytes = int(inout.nbytes) if hasattr(inout, "nbytes") else int(inout.numel() * inout.element_size())
    n = nbytes // _ELEM_BYTES[field]
    if n & (n - 1):
        raise ValueError("inout.len() is not power of 2")       # lib.rs:21-24
    lg = n.bit_length() - 1 if n else 0
    p, _k = ffi.as_pointer(inout)
    if stream is None:
        err = L.compute_ntt(device_id, p, lg, int(order), int(direction), int(ntt_type))
    else:
        err = L.sppark_ntt(device_id, p, lg, int(order), int(direction), int(ntt_type), stream)
    ffi.check(L, err)
    return inout


def NTT(device_id, inout, order, field="gl64", stream=None):
    return compute_ntt(device_id, inout, order, NTTDirection.Forward, NTTType.Standard, field, stream)


def iNTT(device_id, inout, order, field="gl64", stream=None):
    return compute_ntt(device_id, inout, order, NTTDirection.Inverse, NTTType.Standard, field, stream)


def coset_NTT(device_id, inout, order, field="gl64", stream=None):
    return compute_ntt(device_id, inout, order, NTTDirection.Forward, NTTType.Coset, field, stream)


def coset_iNTT(device_id, inout, order, field="gl64", stream=None):
    return compute_ntt(device_id, inout, order, NTTDirection.Inverse, NTTType.Coset, field, stream)
