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
_ELEM_BYTES = {"gl64": 8, "bb31": 4, "bls12_381": 32, "bn254": 32, "bls12_377": 32, "pallas": 32, "vesta": 32, "gl64_plonky2": 8, "bb31_canonical": 4}


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


def _nelems(buf, field):
    nbytes = int(buf.nbytes) if hasattr(buf, "nbytes") else int(buf.numel() * buf.element_size())
    return nbytes // _ELEM_BYTES[field]


def LDE(device_id, inout, lg_domain_size, lg_blowup, field="gl64", aux_out=None, stream=None):
    """NTT::LDE / LDE_aux (ntt/ntt.cuh:283-340), in place: |inout| (host or device) has
    2^(lg_domain_size+lg_blowup) elements, the first 2^lg_domain_size hold the evaluations
    on the small domain; on return: evaluations on the coset of the extended domain, natural
    order.  aux_out (optional, 2^lg_domain_size elements) receives the coefficients."""
    L = ffi.load(field)
    if _nelems(inout, field) != 1 << (lg_domain_size + lg_blowup):
        raise ValueError("inout must hold 2^(lg_domain_size+lg_blowup) elements")
    if aux_out is not None and _nelems(aux_out, field) != 1 << lg_domain_size:
        raise ValueError("aux_out must hold 2^lg_domain_size elements")
    p, _k = ffi.as_pointer(inout)
    a, _k2 = ffi.as_pointer(aux_out) if aux_out is not None else (None, None)
    ffi.check(L, L.sppark_lde(device_id, p, lg_domain_size, lg_blowup, a, stream))
    return inout


def LDE_powers(device_id, d_inout, field="gl64", stream=None):
    """NTT::LDE_powers(stream, d_inout, lg) (ntt/ntt.cuh:352-356): d_inout[i] *= g^bitrev(i); device buffer."""
    L = ffi.load(field)
    n = _nelems(d_inout, field)
    if n & (n - 1) or n == 0:
        raise ValueError("length is not power of 2")
    p, _k = ffi.as_pointer(d_inout)
    ffi.check(L, L.sppark_lde_powers(device_id, p, n.bit_length() - 1, stream))
    return d_inout


def LDE_expand(device_id, d_out, d_in, lg_domain_size, lg_blowup, field="gl64", stream=None):
    """NTT::LDE_expand (ntt/ntt.cuh:358-365): d_out[i << lg_blowup] = d_in[i], zeros elsewhere; device buffers."""
    L = ffi.load(field)
    if _nelems(d_in, field) != 1 << lg_domain_size or _nelems(d_out, field) != 1 << (lg_domain_size + lg_blowup):
        raise ValueError("buffer sizes do not match the domain sizes")
    po, _k = ffi.as_pointer(d_out)
    pi, _k2 = ffi.as_pointer(d_in)
    ffi.check(L, L.sppark_lde_expand(device_id, po, pi, lg_domain_size, lg_blowup, stream))
    return d_out
