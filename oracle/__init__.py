"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so, the CPU restatement of the reference's
MSM / NTT hot path (see oracle/ff.hpp, ec.hpp, msm.hpp, ntt.hpp for the
reference file:line each function follows).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under sppark_amd/ does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLS12_381, BN254 = 0, 1
BLS12_381_G2, BN254_G2 = 2, 3          # same curves, group G2 (coordinates in Fp2 = c0 | c1)
BLS12_377, BLS12_377_G2 = 4, 5
PALLAS, VESTA = 6, 7                   # the Pasta cycle: no G2
FIELD_BLS_FP, FIELD_BLS_FR, FIELD_BN_FP, FIELD_BN_FR = 0, 1, 2, 3
FIELD_BLS377_FP, FIELD_BLS377_FR = 4, 5
FIELD_PASTA_P, FIELD_PASTA_Q = 6, 7
CURVE_ID = {"bls12_381": BLS12_381, "bn254": BN254, "bls12_377": BLS12_377, "pallas": PALLAS, "vesta": VESTA}
CURVE_ID_G2 = {"bls12_381": BLS12_381_G2, "bn254": BN254_G2, "bls12_377": BLS12_377_G2}
FP_FIELD_ID = {BLS12_381: FIELD_BLS_FP, BN254: FIELD_BN_FP, BLS12_377: FIELD_BLS377_FP, PALLAS: FIELD_PASTA_P, VESTA: FIELD_PASTA_Q}
FR_FIELD_ID = {BLS12_381: FIELD_BLS_FR, BN254: FIELD_BN_FR, BLS12_377: FIELD_BLS377_FR, PALLAS: FIELD_PASTA_Q, VESTA: FIELD_PASTA_P}
NN, NR, RN, RR = 0, 1, 2, 3
FORWARD, INVERSE = 0, 1
STANDARD, COSET = 0, 1

FP_BYTES = {BLS12_381: 48, BN254: 32, BLS12_381_G2: 96, BN254_G2: 64, BLS12_377: 48, BLS12_377_G2: 96, PALLAS: 32, VESTA: 32}      # bytes per coordinate
FR_MODULUS = {
    BLS12_381: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    BN254: int("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 16),
}
FP_MODULUS = {
    BLS12_381: int("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab", 16),
    BN254: int("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", 16),
}
FR_MODULUS[BLS12_377] = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
FP_MODULUS[BLS12_377] = int("01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001", 16)
FR_MODULUS[BLS12_381_G2] = FR_MODULUS[BLS12_381]; FR_MODULUS[BN254_G2] = FR_MODULUS[BN254]
FP_MODULUS[BLS12_381_G2] = FP_MODULUS[BLS12_381]; FP_MODULUS[BN254_G2] = FP_MODULUS[BN254]
FR_MODULUS[BLS12_377_G2] = FR_MODULUS[BLS12_377]; FP_MODULUS[BLS12_377_G2] = FP_MODULUS[BLS12_377]
FP_MODULUS[PALLAS] = FR_MODULUS[VESTA] = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
FP_MODULUS[VESTA] = FR_MODULUS[PALLAS] = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
GL64_P = 0xffffffff00000001
BB31_P = 0x78000001


def build(force=False):
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force and os.path.exists(so):
        os.remove(so)
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])     # make: a no-op when up to date
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, sz, ci, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        L.oracle_field_op.argtypes = [ci, ci, vp, vp, vp]
        L.oracle_g1_generator.argtypes = [ci, vp, sz]
        L.oracle_g1_gen_points.argtypes = [ci, vp, sz, sz, u64]
        L.oracle_g1_mul.argtypes = [ci, vp, vp, sz, vp]
        L.oracle_g1_on_curve.argtypes = [ci, vp, sz]
        L.oracle_jac_to_affine.argtypes = [ci, vp, vp, sz]
        L.oracle_xyzz_to_affine.argtypes = [ci, vp, vp, sz]
        L.oracle_jac_add.argtypes = [ci, vp, vp, vp]
        L.oracle_jac_dbl.argtypes = [ci, vp, vp]
        L.oracle_jac_eq.argtypes = [ci, vp, vp]
        L.oracle_msm.argtypes = [ci, ci, vp, vp, sz, sz, vp, ci, sz]
        L.oracle_ntt_gl64.argtypes = [vp, ctypes.c_uint, ci, ci, ci]
        L.oracle_ntt_bb31.argtypes = [vp, ctypes.c_uint, ci, ci, ci]
        L.oracle_ntt_naive_gl64.argtypes = [vp, vp, ctypes.c_uint, ci]
        L.oracle_ntt_naive_bb31.argtypes = [vp, vp, ctypes.c_uint, ci]
        L.oracle_ntt_fr.argtypes = [ci, vp, ctypes.c_uint, ci, ci, ci]
        L.oracle_ntt_naive_fr.argtypes = [ci, vp, vp, ctypes.c_uint, ci]
        L.oracle_lde.argtypes = [ci, vp, ctypes.c_uint, ctypes.c_uint, vp]
        L.oracle_lde_powers.argtypes = [ci, vp, ctypes.c_uint]
        L.oracle_lde_expand.argtypes = [ci, vp, vp, ctypes.c_uint, ctypes.c_uint]
        L.oracle_set_root_conventions.argtypes = [ci, ci]
        L.oracle_prefix_op.argtypes = [ci, vp, vp, sz, ci]
        L.oracle_poly_evaluate.argtypes = [ci, vp, vp, sz, vp, sz]
        L.oracle_div_by_x_minus_z.argtypes = [ci, vp, sz, vp, ci]
        L.oracle_fr_root.argtypes = [ci, vp, ctypes.c_uint]
        L.oracle_gl64_root.argtypes = [ctypes.c_uint]; L.oracle_gl64_root.restype = u64
        L.oracle_bb31_root.argtypes = [ctypes.c_uint]; L.oracle_bb31_root.restype = ctypes.c_uint32
        L.oracle_gl64_mul.argtypes = [u64, u64]; L.oracle_gl64_mul.restype = u64
        L.oracle_bb31_mul.argtypes = [ctypes.c_uint32] * 2; L.oracle_bb31_mul.restype = ctypes.c_uint32
        L.oracle_bb31_to_mont.argtypes = [ctypes.c_uint32]; L.oracle_bb31_to_mont.restype = ctypes.c_uint32
        L.oracle_bb31_from_mont.argtypes = [ctypes.c_uint32]; L.oracle_bb31_from_mont.restype = ctypes.c_uint32
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------- fields ----
def int_to_limbs(x, nbytes):
    return np.frombuffer(int(x).to_bytes(nbytes, "little"), dtype=np.uint64).copy()


def limbs_to_int(a):
    return int.from_bytes(np.ascontiguousarray(a).tobytes(), "little")


def field_op(field, op, a, b=None):
    nbytes = 48 if field in (FIELD_BLS_FP, FIELD_BLS377_FP) else 32
    out = np.zeros(nbytes // 8, dtype=np.uint64)
    a = np.ascontiguousarray(a, dtype=np.uint64)
    bp = _ptr(np.ascontiguousarray(b, dtype=np.uint64)) if b is not None else None
    rc = lib().oracle_field_op(field, op, _ptr(out), _ptr(a), bp)
    assert rc == 0
    return out


# --------------------------------------------------------------------- G1 ----
def g1_generator(curve, stride=None):
    stride = stride or 2 * FP_BYTES[curve]
    out = np.zeros(stride, dtype=np.uint8)
    lib().oracle_g1_generator(curve, _ptr(out), stride)
    return out


def g1_gen_points(curve, n, seed, stride=None):
    """n points k_i*G (k_i from splitmix64(seed)) as an (n, stride) uint8 array."""
    stride = stride or 2 * FP_BYTES[curve]
    out = np.zeros((n, stride), dtype=np.uint8)
    lib().oracle_g1_gen_points(curve, _ptr(out), stride, n, seed)
    return out


def g1_mul(curve, point, k):
    point = np.ascontiguousarray(point, dtype=np.uint8)
    out = np.zeros_like(point)
    kb = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    lib().oracle_g1_mul(curve, _ptr(out), _ptr(point), point.size, _ptr(kb))
    return out


def g1_on_curve(curve, point):
    point = np.ascontiguousarray(point, dtype=np.uint8)
    return bool(lib().oracle_g1_on_curve(curve, _ptr(point), point.size))


def jac_to_affine(curve, jac, stride=None):
    stride = stride or 2 * FP_BYTES[curve]
    jac = np.ascontiguousarray(jac, dtype=np.uint8)
    out = np.zeros(stride, dtype=np.uint8)
    lib().oracle_jac_to_affine(curve, _ptr(out), _ptr(jac), stride)
    return out


def xyzz_to_affine(curve, p, stride=None):
    stride = stride or 2 * FP_BYTES[curve]
    p = np.ascontiguousarray(p, dtype=np.uint8)
    out = np.zeros(stride, dtype=np.uint8)
    lib().oracle_xyzz_to_affine(curve, _ptr(out), _ptr(p), stride)
    return out


def jac_add(curve, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
    out = np.zeros_like(a)
    lib().oracle_jac_add(curve, _ptr(out), _ptr(a), _ptr(b))
    return out


def jac_dbl(curve, a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.zeros_like(a)
    lib().oracle_jac_dbl(curve, _ptr(out), _ptr(a))
    return out


def jac_eq(curve, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
    return bool(lib().oracle_jac_eq(curve, _ptr(a), _ptr(b)))


MSM_PIPPENGER_CPU, MSM_NAIVE, MSM_SIGNED = 0, 1, 2


def msm(curve, points, scalars, algo=MSM_PIPPENGER_CPU, param=0, mont=False):
    """points: (n, stride) uint8; scalars: (n, 32) uint8 LE.  Returns Jacobian bytes."""
    points = np.ascontiguousarray(points, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    n = points.shape[0] if points.ndim == 2 else 0
    stride = points.shape[1] if points.ndim == 2 else 2 * FP_BYTES[curve]
    assert scalars.size == n * 32
    out = np.zeros(3 * FP_BYTES[curve], dtype=np.uint8)
    rc = lib().oracle_msm(curve, algo, _ptr(out), _ptr(points), stride, n, _ptr(scalars), int(mont), param)
    assert rc == 0
    return out


def msm_affine(curve, points, scalars, **kw):
    return jac_to_affine(curve, msm(curve, points, scalars, **kw))


# -------------------------------------------------------------------- NTT ----
def ntt_gl64(a, order=NN, direction=FORWARD, type=STANDARD):
    a = np.array(a, dtype=np.uint64)
    lg = int(a.size).bit_length() - 1
    assert a.size == 1 << lg
    lib().oracle_ntt_gl64(_ptr(a), lg, order, direction, type)
    return a


def ntt_bb31(a, order=NN, direction=FORWARD, type=STANDARD):
    a = np.array(a, dtype=np.uint32)
    lg = int(a.size).bit_length() - 1
    assert a.size == 1 << lg
    lib().oracle_ntt_bb31(_ptr(a), lg, order, direction, type)
    return a


def ntt_fr(curve, a, order=NN, direction=FORWARD, type=STANDARD):
    """a: (n, 4) uint64 Montgomery limbs of the curve's scalar field"""
    a = np.array(a, dtype=np.uint64).reshape(-1, 4)
    lg = int(a.shape[0]).bit_length() - 1
    assert a.shape[0] == 1 << lg
    lib().oracle_ntt_fr(curve, _ptr(a), lg, order, direction, type)
    return a


_LDE_FIELDS = {"gl64": (0, np.uint64, 1), "bb31": (1, np.uint32, 1), "bls12_381": (2, np.uint64, 4), "bn254": (3, np.uint64, 4),
               "bls12_377": (4, np.uint64, 4), "pallas": (6, np.uint64, 4), "vesta": (7, np.uint64, 4),
               "m31": (8, np.uint32, 1), "bb31x4": (9, np.uint32, 4)}          # polynomial primitives only


def lde(field, x, lg_blowup, want_aux=False):
    """NTT::LDE_aux (ntt/ntt.cuh:283-336) on 2^lg_domain evaluations x; returns the
    2^(lg_domain+lg_blowup) coset evaluations (and the coefficients when want_aux)."""
    fid, dt, w = _LDE_FIELDS[field]
    x = np.ascontiguousarray(x, dtype=dt).reshape(-1, w)
    dom = x.shape[0]
    lg = dom.bit_length() - 1
    assert dom == 1 << lg
    ext = np.zeros((dom << lg_blowup, w), dtype=dt)
    ext[:dom] = x
    aux = np.zeros((dom, w), dtype=dt) if want_aux else None
    lib().oracle_lde(fid, _ptr(ext), lg, lg_blowup, _ptr(aux) if want_aux else None)
    ext = ext.reshape(-1) if w == 1 else ext
    if want_aux:
        return ext, (aux.reshape(-1) if w == 1 else aux)
    return ext


def lde_powers(field, x):
    fid, dt, w = _LDE_FIELDS[field]
    x = np.array(x, dtype=dt).reshape(-1, w)
    lib().oracle_lde_powers(fid, _ptr(x), x.shape[0].bit_length() - 1)
    return x.reshape(-1) if w == 1 else x


def lde_expand(field, x, lg_blowup):
    fid, dt, w = _LDE_FIELDS[field]
    x = np.ascontiguousarray(x, dtype=dt).reshape(-1, w)
    out = np.zeros((x.shape[0] << lg_blowup, w), dtype=dt)
    lib().oracle_lde_expand(fid, _ptr(out), _ptr(x), x.shape[0].bit_length() - 1, lg_blowup)
    return out.reshape(-1) if w == 1 else out


def set_root_conventions(goldilocks_plonky2=False, baby_bear_canonical=False):
    """the reference's compile-time NTT root conventions (-DGOLDILOCKS_PLONKY2, -DBABY_BEAR_CANONICAL)"""
    lib().oracle_set_root_conventions(int(goldilocks_plonky2), int(baby_bear_canonical))


# ------------------------------------------------- polynomial primitives -----
def _poly_arr(field, x):
    fid, dt, w = _LDE_FIELDS[field]
    return fid, np.ascontiguousarray(x, dtype=dt).reshape(-1, w), w


def prefix_op(field, x, op):
    """inclusive scan; op 0 = Add, 1 = Multiply (polynomial/prefix_op.cuh)"""
    fid, a, w = _poly_arr(field, x)
    out = np.zeros_like(a)
    lib().oracle_prefix_op(fid, _ptr(out), _ptr(a), a.shape[0], op)
    return out.reshape(-1) if w == 1 else out


def poly_evaluate(field, coeffs, xs):
    """ret[j] = sum_i coeffs[i] * xs[j]^i (polynomial/evaluate.cuh)"""
    fid, c, w = _poly_arr(field, coeffs)
    _, x, _ = _poly_arr(field, xs)
    ret = np.zeros_like(x)
    lib().oracle_poly_evaluate(fid, _ptr(ret), _ptr(x), x.shape[0], _ptr(c), c.shape[0])
    return ret.reshape(-1) if w == 1 else ret


def div_by_x_minus_z(field, coeffs, z, rotate=False):
    """synthetic division (polynomial/div_by_x_minus_z.cuh): remainder first / last (rotate)"""
    fid, c, w = _poly_arr(field, coeffs)
    c = c.copy()
    _, zz, _ = _poly_arr(field, z)
    lib().oracle_div_by_x_minus_z(fid, _ptr(c), c.shape[0], _ptr(zz), int(rotate))
    return c.reshape(-1) if w == 1 else c


def ntt_naive_fr(curve, a, inverse=False):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(a)
    lib().oracle_ntt_naive_fr(curve, _ptr(out), _ptr(a), int(a.shape[0]).bit_length() - 1, int(inverse))
    return out


def fr_root(curve, lg):
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_fr_root(curve, _ptr(out), lg)
    return out


def ntt_naive_gl64(a, inverse=False):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros_like(a)
    lib().oracle_ntt_naive_gl64(_ptr(out), _ptr(a), int(a.size).bit_length() - 1, int(inverse))
    return out


def ntt_naive_bb31(a, inverse=False):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.zeros_like(a)
    lib().oracle_ntt_naive_bb31(_ptr(out), _ptr(a), int(a.size).bit_length() - 1, int(inverse))
    return out


# ------------------------------------------------------------- test inputs ---
def random_scalars(curve, n, seed):
    """n uniform scalars in [0, r) as an (n, 32) uint8 LE array (plain form)."""
    rng = np.random.default_rng(seed)
    r = FR_MODULUS[curve]
    raw = rng.integers(0, 256, size=(n, 40), dtype=np.uint8)
    out = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        out[i] = np.frombuffer((int.from_bytes(raw[i].tobytes(), "little") % r).to_bytes(32, "little"), dtype=np.uint8)
    return out


# ------------------------------------------------ the reference itself (_ref) -
_REF = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_msm.so"))


def ref_msm_affine(curve, points, scalars, nthreads=0, mont=False):
    """The reference's own msm/pippenger.hpp (oracle/_ref/libref_msm.so); affine X|Y."""
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_msm.so"))
        _REF.ref_mult_pippenger.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
    points = np.ascontiguousarray(points, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    out = np.zeros(2 * FP_BYTES[curve], dtype=np.uint8)
    rc = _REF.ref_mult_pippenger(curve, _ptr(out), _ptr(points), points.shape[1], points.shape[0],
                                 _ptr(scalars), int(mont), nthreads)
    assert rc == 0
    return out


# ------------------- the reference's own NTT, built for gfx950 by its HIP path (_ref/libref_ntt_*.so) -
# oracle/Makefile: ref_ntt; oracle/ref_ntt_shim.cu.  GPU box only (the libraries hold device code); used by the `-m gpu`
# tests and by bench.py's reference leg, never by the product.
_REF_NTT = {}
REF_NTT_FIELDS = ("gl64", "gl64_plonky2", "bb31", "bb31_canonical", "bls12_381", "bn254", "bls12_377", "pallas", "vesta")


class _RefError(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_char_p)]      # RustError::by_value (util/rusterror.h:18-36)


def ref_ntt_available(field):
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_ntt_%s.so" % field))


def ref_ntt_lib(field):
    if field not in _REF_NTT:
        L = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_ntt_%s.so" % field), mode=ctypes.RTLD_LOCAL)
        vp, u32, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        L.compute_ntt.argtypes = [ctypes.c_size_t, vp, u32, ci, ci, ci]
        L.compute_ntt.restype = _RefError
        L.ref_ntt_dev.argtypes = [vp, u32, ci, ci, ci]
        L.ref_ntt_dev_timed.argtypes = [vp, u32, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_float)]
        L.ref_lde.argtypes = [vp, u32, u32]
        L.ref_lde_aux.argtypes = [vp, u32, u32, vp]
        L.ref_ntt_elem_bytes.restype = ctypes.c_size_t
        _REF_NTT[field] = L
    return _REF_NTT[field]


def ref_compute_ntt(field, x, order, direction, typ):
    """the reference's compute_ntt (poc/ntt-cuda/cuda/ntt_api.cu:25-36) on a copy of the host array |x|"""
    L = ref_ntt_lib(field)
    y = np.ascontiguousarray(x).copy()
    n = y.nbytes // L.ref_ntt_elem_bytes()
    assert n and not n & (n - 1)
    err = L.compute_ntt(0, _ptr(y), n.bit_length() - 1, int(order), int(direction), int(typ))
    assert err.code == 0, (err.code, err.message)
    return y


def ref_ntt_dev(field, dev_ptr, lg, order, direction, typ):
    """NTT::Base_dev_ptr (ntt/ntt.cuh:344-350) on caller-owned device memory, synchronous"""
    rc = ref_ntt_lib(field).ref_ntt_dev(dev_ptr, lg, int(order), int(direction), int(typ))
    assert rc == 0, rc


def ref_ntt_dev_ms(field, dev_ptr, lg, order, direction, typ, iters):
    """average milliseconds of one of |iters| back-to-back reference transforms of a device buffer (event-timed)"""
    ms = ctypes.c_float(0)
    rc = ref_ntt_lib(field).ref_ntt_dev_timed(dev_ptr, lg, int(order), int(direction), int(typ), iters, ctypes.byref(ms))
    assert rc == 0, rc
    return float(ms.value)


def ref_lde(field, x, lg_blowup, want_aux=False):
    """NTT::LDE / LDE_aux (ntt/ntt.cuh:280-342) of the host array |x| (2^lg elements)"""
    L = ref_ntt_lib(field)
    x = np.ascontiguousarray(x)
    eb = L.ref_ntt_elem_bytes()
    n = x.nbytes // eb
    lg = n.bit_length() - 1
    buf = np.zeros((n << lg_blowup) * eb, dtype=np.uint8)
    buf[:n * eb] = x.view(np.uint8).reshape(-1)
    if want_aux:
        aux = np.zeros(n * eb, dtype=np.uint8)
        assert L.ref_lde_aux(_ptr(buf), lg, lg_blowup, _ptr(aux)) == 0
        return buf.view(x.dtype), aux.view(x.dtype)
    assert L.ref_lde(_ptr(buf), lg, lg_blowup) == 0
    return buf.view(x.dtype)


# ------------- the reference's own device field classes fp_t / fr_t (ff/mont_t.hip), _ref/libref_field_<curve>.so -
# oracle/Makefile: ref_field; oracle/ref_field_shim.cu.  GPU box only; the only reference-held pin of the MSM side.
_REF_FIELD = {}
REF_FIELD_CURVES = ("bls12_381", "bn254", "bls12_377", "pallas", "vesta")


def ref_field_available(curve):
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_field_%s.so" % curve))


def ref_field_op(curve, field, op, a, b=None):
    """element-wise op of the reference's fp_t (field 0) / fr_t (field 1) on the GPU (ff/mont_t.hip:96-218):
    0 a+b, 1 a-b, 2 a*b, 3 sqr, 4 to(), 5 from().  |a|, |b|: uint8 arrays of n elements in the reference's memory image."""
    if curve not in _REF_FIELD:
        L = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_field_%s.so" % curve), mode=ctypes.RTLD_LOCAL)
        L.ref_field_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.ref_field_bytes.argtypes = [ctypes.c_int]; L.ref_field_bytes.restype = ctypes.c_size_t
        _REF_FIELD[curve] = L
    L = _REF_FIELD[curve]
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    b = a if b is None else np.ascontiguousarray(b).view(np.uint8).reshape(-1)
    eb = L.ref_field_bytes(field)
    assert a.nbytes % eb == 0 and a.nbytes == b.nbytes
    out = np.zeros_like(a)
    rc = L.ref_field_op(field, op, _ptr(out), _ptr(a), _ptr(b), a.nbytes // eb)
    assert rc == 0, rc
    return out
