"""ctypes binding of the sppark_amd C ABI (include/sppark_amd.h).

What a maintainer of the reference's callers would write:
  * Rust  `#[repr(C)] struct Error { code: i32, str: Option<NonNull<c_char>> }`
    returned by value (rust/src/lib.rs:9-23)               -> class _Error below
  * Go    dlopen the library next to the executable and dlsym each entry point
    (go/sppark.go:83-96,165-214)                           -> load() below
Errors surface as SpparkError (the Rust wrappers `panic!(String::from(err))`,
poc/msm-cuda/src/lib.rs:76-78).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

CURVES = ("bls12_381", "bn254", "bls12_377", "pallas", "vesta")
POLY_ONLY = ("m31", "bb31x4")                        # field types without NTT parameters: polynomial primitives only
NO_G2 = ("pallas", "vesta")                          # no pairing: no mult_pippenger_fp2_inf / sppark_g2_*
NTT_FIELDS = ("gl64", "bb31", "gl64_plonky2", "bb31_canonical")      # the last two: root-convention variants


class _Error(ctypes.Structure):                 # util/rusterror.h:18-36
    _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_void_p)]


class SpparkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("sppark error %d: %s" % (code, message))
        self.code = code
        self.message = message


def lib_path(name):
    # (SPPARK_LIBDIR: a second build with other flags for an A/B job of tools/, see sppark_amd/build.py)
    return os.path.join(_HERE, os.environ.get("SPPARK_LIBDIR", "lib"), "libsppark_%s.so" % name)


def load(name):
    """dlopen libsppark_<name>.so and declare its entry points.  Fails loudly
    when the HIP library has not been built -- there is no fallback."""
    if name in _LIBS:
        return _LIBS[name]
    # torch wheels bundle their own HIP runtime; if /opt/rocm's copy gets loaded
    # first (as a dependency of our library) torch later reports no GPU.  Import
    # torch first so that one runtime serves both.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path(name)
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s is missing: build the HIP libraries first (python -m sppark_amd.build)" % path)
    L = ctypes.CDLL(path)
    vp, sz, ci, cu = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint
    L.cuda_available.restype = ctypes.c_bool
    L.drop_error_message.argtypes = [vp]
    L.cuda_func.argtypes = [vp]
    L.cuda_func.restype = _Error
    L.drop_gpu_ptr_t.argtypes = [ctypes.POINTER(vp)]
    L.clone_gpu_ptr_t.argtypes = [ctypes.POINTER(vp)]
    L.clone_gpu_ptr_t.restype = vp
    if name in CURVES:
        L.mult_pippenger_inf.argtypes = [vp, vp, sz, vp, sz]
        L.mult_pippenger_inf.restype = _Error
        if name not in NO_G2:
            L.mult_pippenger_fp2_inf.argtypes = [vp, vp, sz, vp, sz]
            L.mult_pippenger_fp2_inf.restype = _Error
            L.sppark_msm_g2_path.argtypes = [ctypes.c_uint]
            L.sppark_msm_g2_path.restype = _Error
            L.sppark_g2_jacobian_sum.argtypes = [vp, vp, sz]
            L.sppark_g2_jacobian_sum.restype = None
            L.sppark_g2_to_affine.argtypes = [vp, vp]
            L.sppark_g2_to_affine.restype = None
        L.mult_pippenger.argtypes = [vp, vp, sz, vp]
        L.mult_pippenger.restype = _Error
        L.sppark_msm_create.argtypes = [ctypes.POINTER(vp), ci, vp]
        L.sppark_msm_create.restype = _Error
        L.sppark_msm_destroy.argtypes = [vp]
        L.sppark_msm_set_stream.argtypes = [vp, vp]
        L.sppark_msm_set_stream.restype = _Error
        L.sppark_msm_tune.argtypes = [vp, cu, cu, cu, cu, cu]
        L.sppark_msm_tune.restype = _Error
        L.sppark_msm_release_cached.argtypes = []
        L.sppark_msm_release_cached.restype = None
        L.sppark_msm_tune_split.argtypes = [vp, cu]
        L.sppark_msm_tune_split.restype = _Error
        L.sppark_msm_tune_sums.argtypes = [vp, cu]
        L.sppark_msm_tune_sums.restype = _Error
        L.sppark_msm_tune_sort.argtypes = [vp, cu]
        L.sppark_msm_tune_sort.restype = _Error
        L.sppark_msm_tune_tail.argtypes = [vp, cu, cu]
        L.sppark_msm_tune_tail.restype = _Error
        L.sppark_msm_tune_pipeline.argtypes = [vp, cu, sz, sz]
        L.sppark_msm_tune_pipeline.restype = _Error
        L.sppark_msm_last_chunks.argtypes = [vp]
        L.sppark_msm_last_chunks.restype = cu
        L.sppark_msm_plan_sort.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_uint * 8)]
        L.sppark_msm_tune_records.argtypes = [vp, cu]
        L.sppark_msm_tune_records.restype = _Error
        L.sppark_msm_tail_redone.argtypes = [vp]
        L.sppark_msm_tail_redone.restype = cu
        L.sppark_msm_plan_groups.argtypes = [vp, sz]
        L.sppark_msm_plan_groups.restype = cu
        L.sppark_batch_addition.argtypes = [vp, vp, sz, vp, vp, sz]
        L.sppark_batch_addition.restype = _Error
        L.sppark_ngpus.argtypes = []
        L.sppark_ngpus.restype = sz
        L.sppark_msm_multi.argtypes = [vp, vp, sz, vp, ci, sz, cu]
        L.sppark_msm_multi.restype = _Error
        L.sppark_msm_multi_shards.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), ci, sz, cu,
                                              ctypes.POINTER(ci)]
        L.sppark_msm_multi_shards.restype = _Error
        L.sppark_msm_multi_ms.argtypes = [vp, vp, sz, vp, ci, sz, cu, ctypes.POINTER(ctypes.c_float)]
        L.sppark_msm_multi_ms.restype = _Error
        L.sppark_msm_multi_shards_ms.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), ci, sz, cu,
                                                 ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_float)]
        L.sppark_msm_multi_shards_ms.restype = _Error
        L.sppark_msm_rccl_sum.argtypes = [vp, vp, ci, vp, vp]
        L.sppark_msm_rccl_sum.restype = _Error
        L.sppark_msm_rccl.argtypes = [vp, vp, sz, vp, ci, sz, vp, vp]
        L.sppark_msm_rccl.restype = _Error
        L.sppark_msm_reserve.argtypes = [vp, sz, sz, ci, ci]
        L.sppark_msm_reserve.restype = _Error
        L.sppark_msm_invoke.argtypes = [vp, vp, vp, sz, vp, ci, sz]
        L.sppark_msm_invoke.restype = _Error
        L.sppark_msm_set_points.argtypes = [vp, vp, sz, sz]
        L.sppark_msm_set_points.restype = _Error
        L.sppark_msm_set_points_fixed_base.argtypes = [vp, vp, sz, sz]
        L.sppark_msm_set_points_fixed_base.restype = _Error
        L.sppark_msm_fixed_base_windows.argtypes = [vp]
        L.sppark_msm_fixed_base_windows.restype = cu
        L.sppark_msm_preloaded.argtypes = [vp]
        L.sppark_msm_preloaded.restype = sz
        L.sppark_msm_enable_timing.argtypes = [vp, ci]
        L.sppark_msm_enable_timing.restype = _Error
        L.sppark_msm_kernel_ms.argtypes = [vp, ci]
        L.sppark_msm_kernel_ms.restype = ctypes.c_float
        L.sppark_msm_scratch_bytes.argtypes = [vp]
        L.sppark_msm_scratch_bytes.restype = sz
        L.sppark_msm_plan.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_uint * 8)]
        L.sppark_g1_jacobian_sum.argtypes = [vp, vp, sz]
        L.sppark_g1_to_affine.argtypes = [vp, vp]
        L.sppark_g1_generate.argtypes = [vp, sz, sz, ctypes.c_uint64]
        L.sppark_g1_generate.restype = _Error
        L.sppark_g1_generate_progression.argtypes = [vp, sz, sz, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.sppark_g1_generate_progression.restype = _Error
    if name in NTT_FIELDS or name in CURVES:
        L.compute_ntt.argtypes = [sz, vp, ctypes.c_uint32, ci, ci, ci]
        L.compute_ntt.restype = _Error
        L.sppark_ntt.argtypes = [sz, vp, ctypes.c_uint32, ci, ci, ci, vp]
        L.sppark_ntt.restype = _Error
        u32 = ctypes.c_uint32
        L.sppark_lde.argtypes = [sz, vp, u32, u32, vp, vp]
        L.sppark_lde.restype = _Error
        L.sppark_lde_powers.argtypes = [sz, vp, u32, vp]
        L.sppark_lde_powers.restype = _Error
        L.sppark_lde_expand.argtypes = [sz, vp, vp, u32, u32, vp]
        L.sppark_lde_expand.restype = _Error
        L.sppark_ntt_release_cached.argtypes = []
        L.sppark_ntt_release_cached.restype = None
        L.sppark_ntt_cached_scratch_bytes.argtypes = []
        L.sppark_ntt_cached_scratch_bytes.restype = sz
        L.sppark_ntt_cached_tables.argtypes = []
        L.sppark_ntt_cached_tables.restype = sz

    if name in NTT_FIELDS or name in CURVES or name in POLY_ONLY:
        L.sppark_prefix_op.argtypes = [sz, vp, vp, sz, ci, vp]
        L.sppark_prefix_op.restype = _Error
        L.sppark_poly_evaluate.argtypes = [sz, vp, vp, sz, vp, sz, vp]
        L.sppark_poly_evaluate.restype = _Error
        L.sppark_div_by_x_minus_z.argtypes = [sz, vp, sz, vp, ci, vp]
        L.sppark_div_by_x_minus_z.restype = _Error
    _LIBS[name] = L
    return L


def load_devtest(name):
    """dlopen libsppark_<name>_devtest.so: the device test hooks (sppark_devtest_*), a TEST-only
    library built next to the product ones by sppark_amd.build."""
    key = name + "_devtest"
    if key in _LIBS:
        return _LIBS[key]
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path(key)
    if not os.path.exists(path):
        raise FileNotFoundError("%s is missing: python -m sppark_amd.build" % path)
    L = ctypes.CDLL(path)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.drop_error_message.argtypes = [vp]
    if name in CURVES:
        L.sppark_devtest_field_op.argtypes = [ci, ci, vp, vp, vp, sz]
        L.sppark_devtest_field_op.restype = _Error
        L.sppark_devtest_bucket_field_op.argtypes = [ci, vp, vp, vp, sz]
        L.sppark_devtest_bucket_field_op.restype = _Error
        L.sppark_devtest_bucket_field_limbs.argtypes = []
        L.sppark_devtest_bucket_field_limbs.restype = ci
        L.sppark_devtest_bucket_xyzz_op.argtypes = [ci, vp, vp, vp, sz]
        L.sppark_devtest_bucket_xyzz_op.restype = _Error
        L.sppark_devtest_xyzz_op.argtypes = [ci, vp, vp, vp, sz]
        L.sppark_devtest_xyzz_op.restype = _Error
    else:
        L.sppark_devtest_small_field_op.argtypes = [ci, vp, vp, vp, sz]
        L.sppark_devtest_small_field_op.restype = _Error
    _LIBS[key] = L
    return L


def check(L, err):
    """Raise SpparkError for a non-zero code; take ownership of the message."""
    if err.code != 0:
        msg = ""
        if err.message:
            msg = ctypes.string_at(err.message).decode(errors="replace")
            L.drop_error_message(err.message)
        raise SpparkError(err.code, msg)


def cuda_available(name="bls12_381"):
    """util/all_gpus.cpp:65-66 via go/sppark.go:475-477 IsCudaAvailable()."""
    return bool(load(name).cuda_available())


def as_pointer(x):
    """(address, keepalive) of a numpy array, torch tensor (host or device) or int."""
    if isinstance(x, int):
        return x, None
    if hasattr(x, "data_ptr"):                      # torch tensor
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr(), x
    if hasattr(x, "ctypes"):                        # numpy
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data, x
    raise TypeError("unsupported buffer type %r" % type(x))
