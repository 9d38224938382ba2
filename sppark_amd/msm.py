"""MSM host API: mirror of poc/msm-cuda/src/lib.rs:18-119 (multi_scalar_mult,
multi_scalar_mult_arkworks) plus a context object playing the role of the
reference's C++-only msm_t (msm/pippenger.cuh:325-728): scratch memory reused
across calls, device-resident inputs, stream selection and kernel timers.
"""
import ctypes

import numpy as np

from . import ffi

FP_BYTES = {"bls12_381": 48, "bn254": 32, "bls12_377": 48, "pallas": 32, "vesta": 32}


def _nbytes(x):
    if hasattr(x, "nbytes"):
        return int(x.nbytes)
    return int(x.numel() * x.element_size())


def _npoints(points, stride):
    total = _nbytes(points)
    if total % stride:
        raise ValueError("points buffer is not a multiple of the %d-byte stride" % stride)
    return total // stride


def multi_scalar_mult_arkworks(points, scalars, curve="bls12_381", ffi_affine_sz=None):
    """mult_pippenger_inf (poc/msm-cuda/src/lib.rs:46-82).

    points : buffer of Affine_inf_t records (X | Y | infinity flag), stride
             ffi_affine_sz (default: 2*sizeof(fp) + 8, the arkworks layout)
    scalars: n 32-byte little-endian integers < r (not Montgomery)
    returns the Jacobian result X|Y|Z as a uint8 array (144 B / 96 B).
    Buffers may be numpy arrays or torch tensors (host or device).
    """
    L = ffi.load(curve)
    fb = FP_BYTES[curve]
    stride = ffi_affine_sz or (2 * fb + 8)
    n = _npoints(points, stride)
    if _nbytes(scalars) != 32 * n:
        raise ValueError("length mismatch")                 # lib.rs:61-63
    pp, _k1 = ffi.as_pointer(points)
    sp, _k2 = ffi.as_pointer(scalars)
    out = np.zeros(3 * fb, dtype=np.uint8)
    ffi.check(L, L.mult_pippenger_inf(out.ctypes.data, pp, n, sp, stride))
    return out


def multi_scalar_mult(points, scalars, curve="bls12_381"):
    """mult_pippenger (poc/msm-cuda/src/lib.rs:18-44): plain X|Y affine points,
    infinity encoded as all-zero."""
    L = ffi.load(curve)
    fb = FP_BYTES[curve]
    n = _npoints(points, 2 * fb)
    if _nbytes(scalars) != 32 * n:
        raise ValueError("length mismatch")
    pp, _k1 = ffi.as_pointer(points)
    sp, _k2 = ffi.as_pointer(scalars)
    out = np.zeros(3 * fb, dtype=np.uint8)
    ffi.check(L, L.mult_pippenger(out.ctypes.data, pp, n, sp))
    return out


class MsmContext:
    """Reusable MSM context (cf. msm_t, msm/pippenger.cuh:325-388,582-610)."""

    def __init__(self, curve="bls12_381", device_id=-1, stream=None):
        self.curve = curve
        self.L = ffi.load(curve)
        self.fb = FP_BYTES[curve]
        h = ctypes.c_void_p()
        ffi.check(self.L, self.L.sppark_msm_create(ctypes.byref(h), device_id, stream))
        self.h = h

    def close(self):
        if self.h:
            self.L.sppark_msm_destroy(self.h)
            self.h = None

    __del__ = close

    def set_stream(self, stream):
        ffi.check(self.L, self.L.sppark_msm_set_stream(self.h, stream))

    def tune(self, wbits=0, L=0, F=0, K=0, nslabs=0):
        ffi.check(self.L, self.L.sppark_msm_tune(self.h, wbits, L, F, K, nslabs))

    def tune_sort(self, low_bits=0):
        ffi.check(self.L, self.L.sppark_msm_tune_sort(self.h, low_bits))

    def tune_split(self, big_partition=0):
        ffi.check(self.L, self.L.sppark_msm_tune_split(self.h, big_partition))

    def tune_sums(self, top_items=0):
        """bucket sums: windows with at most this many partial sums use the subset-sum top (0 = automatic, 1 = never)"""
        ffi.check(self.L, self.L.sppark_msm_tune_sums(self.h, top_items))

    def tune_tail(self, join=0, k1=0):
        """join=1: no k_join_runs (every record segment through the fan-in tree); k1: buckets per work
        item of the first bucket-sum level (0 = as the other levels)"""
        ffi.check(self.L, self.L.sppark_msm_tune_tail(self.h, join, k1))

    def tune_pipeline(self, groups=0, chunk_points=0, max_scratch_bytes=0):
        """window groups (sort of group g+1 under the accumulation of group g), points per chunk
        of the chunked path, upper bound of the scratch memory; 0 = automatic"""
        ffi.check(self.L, self.L.sppark_msm_tune_pipeline(self.h, groups, chunk_points, max_scratch_bytes))

    def last_chunks(self):
        return int(self.L.sppark_msm_last_chunks(self.h))

    def tail_redone(self):
        """invocations whose tail ran twice: the piece tree of a small MSM met a bucket beyond its size (skewed scalars)"""
        return int(self.L.sppark_msm_tail_redone(self.h))

    def plan_groups(self, npoints):
        return int(self.L.sppark_msm_plan_groups(self.h, npoints))

    def reserve(self, npoints, ffi_affine_sz, host_points=False, host_scalars=False):
        ffi.check(self.L, self.L.sppark_msm_reserve(self.h, npoints, ffi_affine_sz,
                                                    int(host_points), int(host_scalars)))

    def enable_timing(self, on=True):
        ffi.check(self.L, self.L.sppark_msm_enable_timing(self.h, int(on)))

    def kernel_ms(self, which):
        return float(self.L.sppark_msm_kernel_ms(self.h, which))

    def scratch_bytes(self):
        return int(self.L.sppark_msm_scratch_bytes(self.h))

    def plan(self, npoints):
        out = (ctypes.c_uint * 8)()
        self.L.sppark_msm_plan(self.h, npoints, ctypes.byref(out))
        keys = ("window_bits", "windows", "buckets_per_window", "run_length", "partitions", "low_bits", "fan_in", "bucket_chunk")
        d = dict(zip(keys, [int(v) for v in out]))
        self.L.sppark_msm_plan_sort(self.h, npoints, ctypes.byref(out))
        keys = ("slabs", "slab_points", "record_index_bits", "lg_slabs_per_group", "index_groups", "window_groups", "first_bucket_chunk", "piece_tree_max")
        d.update(zip(keys, [int(v) for v in out]))
        d["packed_records"] = d["record_index_bits"] != 0
        return d

    def tune_records(self, records=0):
        """level-A sort records: 0 = automatic (4 bytes unless tune(nslabs=...) is given), 1 = 8 bytes, 2 = 4 bytes also with a
        given slab count"""
        ffi.check(self.L, self.L.sppark_msm_tune_records(self.h, records))

    def set_points(self, points, npoints=None, ffi_affine_sz=None, fixed_base=False):
        """Keep a copy of the bases in HBM (msm_t(points, np, ffi_affine_sz),
        msm/pippenger.cuh:351-385); invoke(None, scalars) then uses them.
        fixed_base: also build the per-window multiples of every point (include/sppark_amd.h,
        sppark_msm_set_points_fixed_base); invoke(None, scalars) over all of them is then one window."""
        stride = ffi_affine_sz or 2 * self.fb
        if points is None:
            ffi.check(self.L, self.L.sppark_msm_set_points(self.h, None, 0, stride))
            return
        n = npoints if npoints is not None else _npoints(points, stride)
        pp, _k = ffi.as_pointer(points)
        fn = self.L.sppark_msm_set_points_fixed_base if fixed_base else self.L.sppark_msm_set_points
        ffi.check(self.L, fn(self.h, pp, n, stride))

    def fixed_base_windows(self):
        return int(self.L.sppark_msm_fixed_base_windows(self.h))

    def preloaded(self):
        return int(self.L.sppark_msm_preloaded(self.h))

    def invoke(self, points, scalars, npoints=None, mont=False, ffi_affine_sz=None):
        """points=None: the preloaded bases (invoke(out, scalars), pippenger.cuh:604-605)."""
        stride = ffi_affine_sz or 2 * self.fb
        if points is None:
            n = npoints if npoints is not None else _npoints(scalars, 32)
            pp, _k1 = None, None
        else:
            n = npoints if npoints is not None else _npoints(points, stride)
            pp, _k1 = ffi.as_pointer(points)
        sp, _k2 = ffi.as_pointer(scalars)
        out = np.zeros(3 * self.fb, dtype=np.uint8)
        ffi.check(self.L, self.L.sppark_msm_invoke(self.h, out.ctypes.data, pp, n, sp, int(mont), stride))
        return out


def ngpus(curve="bls12_381"):
    """usable devices (ngpus(), util/all_gpus.cpp:62-63)"""
    return int(ffi.load(curve).sppark_ngpus())


def msm_multi(points, scalars, curve="bls12_381", ndev=0, mont=False, ffi_affine_sz=None, timings=False):
    """sppark_msm_multi: one process, one host thread + context per device, contiguous shards of
    HOST-resident inputs, partial sums added on the host.  Returns the Jacobian result; with
    timings=True (sppark_msm_multi_ms) also the wall-clock milliseconds of every device."""
    L = ffi.load(curve)
    fb = FP_BYTES[curve]
    stride = ffi_affine_sz or 2 * fb
    n = _npoints(points, stride)
    if _nbytes(scalars) != 32 * n:
        raise ValueError("length mismatch")
    pp, _k1 = ffi.as_pointer(points)
    sp, _k2 = ffi.as_pointer(scalars)
    out = np.zeros(3 * fb, dtype=np.uint8)
    if timings:
        ms = (ctypes.c_float * (ndev or int(L.sppark_ngpus())))()
        ffi.check(L, L.sppark_msm_multi_ms(out.ctypes.data, pp, n, sp, int(mont), stride, ndev, ms))
        return out, [float(v) for v in ms]
    ffi.check(L, L.sppark_msm_multi(out.ctypes.data, pp, n, sp, int(mont), stride, ndev))
    return out


def msm_multi_shards(shards, curve="bls12_381", device_ids=None, mont=False, ffi_affine_sz=None, timings=False):
    """sppark_msm_multi_shards: |shards| = [(points, scalars), ...], shard i on device
    device_ids[i] (default: device i); buffers may be host arrays or tensors on that device."""
    L = ffi.load(curve)
    fb = FP_BYTES[curve]
    stride = ffi_affine_sz or 2 * fb
    k = len(shards)
    P = (ctypes.c_void_p * k)(); S = (ctypes.c_void_p * k)(); N = (ctypes.c_size_t * k)()
    keep = []
    for i, (pts, sc) in enumerate(shards):
        n = _npoints(pts, stride) if _nbytes(pts) else 0
        if _nbytes(sc) != 32 * n:
            raise ValueError("length mismatch in shard %d" % i)
        pp, k1 = ffi.as_pointer(pts); sp, k2 = ffi.as_pointer(sc)
        keep += [k1, k2]
        P[i] = pp; S[i] = sp; N[i] = n
    ids = (ctypes.c_int * k)(*device_ids) if device_ids is not None else None
    out = np.zeros(3 * fb, dtype=np.uint8)
    if timings:
        ms = (ctypes.c_float * max(1, k))()
        ffi.check(L, L.sppark_msm_multi_shards_ms(out.ctypes.data, P, N, S, int(mont), stride, k, ids, ms))
        return out, [float(v) for v in ms][:k]
    ffi.check(L, L.sppark_msm_multi_shards(out.ctypes.data, P, N, S, int(mont), stride, k, ids))
    return out


def batch_addition(points, bitmap, refmap=None, curve="bls12_381", ffi_affine_sz=None):
    """sppark_batch_addition (msm/batch_addition.cuh:25-132): sum of the points whose bit is set in
    |bitmap| (uint32 words); with |refmap|: points of bitmap XOR refmap, those only in refmap
    subtracted.  Returns the Jacobian sum."""
    L = ffi.load(curve)
    fb = FP_BYTES[curve]
    stride = ffi_affine_sz or 2 * fb
    n = _npoints(points, stride)
    words = (n + 31) // 32
    if _nbytes(bitmap) != 4 * words or (refmap is not None and _nbytes(refmap) != 4 * words):
        raise ValueError("maps must hold ceil(npoints/32) 32-bit words")
    pp, _k1 = ffi.as_pointer(points)
    pb, _k2 = ffi.as_pointer(bitmap)
    pr, _k3 = ffi.as_pointer(refmap) if refmap is not None else (None, None)
    out = np.zeros(3 * fb, dtype=np.uint8)
    ffi.check(L, L.sppark_batch_addition(out.ctypes.data, pp, n, pb, pr, stride))
    return out


def set_g2_path(mode, curve="bls12_381"):
    """sppark_msm_g2_path: the accumulation kernel of the G2 entry point (process-wide per library): 0 = automatic
    (a pair of waves per addition, one Fp2 component each, for the 14-limb base fields), 1 = wave pairs, 2 = one lane.
    A test / tuning hook: not thread-safe against G2 calls in flight (include/sppark_amd.h)."""
    if curve in ffi.NO_G2:
        raise ffi.SpparkError(-1, "%s has no G2 (no pairing): no sppark_msm_g2_path" % curve)
    L = ffi.load(curve)
    ffi.check(L, L.sppark_msm_g2_path(int(mode)))


def multi_scalar_mult_fp2_arkworks(points, scalars, curve="bls12_381", ffi_affine_sz=None):
    """mult_pippenger_fp2_inf (poc/msm-cuda/src/lib.rs:84-119): MSM over G2.

    points : Affine_inf_t records over Fp2: X.c0|X.c1|Y.c0|Y.c1|infinity flag, stride
             ffi_affine_sz (default 4*sizeof(fp) + 8, the arkworks G2Affine layout)
    returns: Jacobian X|Y|Z, each an Fp2 element (288 B BLS12-381, 192 B bn254)"""
    L = ffi.load(curve)
    fb = FP_BYTES[curve]
    stride = ffi_affine_sz or 4 * fb + 8
    n = _npoints(points, stride)
    if _nbytes(scalars) != 32 * n:
        raise ValueError("length mismatch")
    pp, _k1 = ffi.as_pointer(points)
    sp, _k2 = ffi.as_pointer(scalars)
    out = np.zeros(6 * fb, dtype=np.uint8)
    ffi.check(L, L.mult_pippenger_fp2_inf(out.ctypes.data, pp, n, sp, stride))
    return out


def jacobian_sum_g2(points, curve="bls12_381"):
    L = ffi.load(curve)
    pts = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 6 * FP_BYTES[curve])
    out = np.zeros(6 * FP_BYTES[curve], dtype=np.uint8)
    L.sppark_g2_jacobian_sum(out.ctypes.data, pts.ctypes.data, pts.shape[0])
    return out


def to_affine_g2(jacobian, curve="bls12_381"):
    L = ffi.load(curve)
    j = np.ascontiguousarray(jacobian, dtype=np.uint8)
    out = np.zeros(4 * FP_BYTES[curve], dtype=np.uint8)
    L.sppark_g2_to_affine(out.ctypes.data, j.ctypes.data)
    return out


def jacobian_sum(points, curve="bls12_381"):
    """Sum of Jacobian points (host arithmetic; the multi-GPU combine step)."""
    L = ffi.load(curve)
    pts = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 3 * FP_BYTES[curve])
    out = np.zeros(3 * FP_BYTES[curve], dtype=np.uint8)
    L.sppark_g1_jacobian_sum(out.ctypes.data, pts.ctypes.data, pts.shape[0])
    return out


def to_affine(jacobian, curve="bls12_381"):
    """(x | y) of a Jacobian point, infinity -> all-zero (what the reference's
    tests obtain from arkworks' into_affine(), poc/msm-cuda/tests/msm.rs:26-38)."""
    L = ffi.load(curve)
    j = np.ascontiguousarray(jacobian, dtype=np.uint8)
    out = np.zeros(2 * FP_BYTES[curve], dtype=np.uint8)
    L.sppark_g1_to_affine(out.ctypes.data, j.ctypes.data)
    return out


def generate_points(out, n, seed, stride, curve="bls12_381"):
    """Fill |out| (numpy array or torch tensor, host or device) with n points
    k_i*G computed on the GPU (synthetic inputs in the shape of
    poc/msm-cuda/src/util.rs:11-38)."""
    L = ffi.load(curve)
    p, _k = ffi.as_pointer(out)
    ffi.check(L, L.sppark_g1_generate(p, stride, n, seed))
    return out


def generate_progression(out, n, a, b, stride, curve="bls12_381"):
    """Fill the DEVICE tensor |out| with the n DISTINCT points P_i = (a + i*b)*G (affine, |stride| bytes apart),
    computed and normalised on the GPU: an MSM over them equals (sum s_i (a + i b) mod r)*G, which is how a
    2^26-point result on points that do not repeat is checked (poc/msm-cuda/tests/msm.rs:19-39 uses an
    arbitrary-point oracle; none finishes at that size)."""
    import ctypes
    L = ffi.load(curve)
    p, _k = ffi.as_pointer(out)
    A = (ctypes.c_uint64 * 2)(a & (2**64 - 1), a >> 64)
    B = (ctypes.c_uint64 * 2)(b & (2**64 - 1), b >> 64)
    ffi.check(L, L.sppark_g1_generate_progression(p, stride, n, A, B))
    return out
