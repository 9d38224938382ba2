"""CPU: the C-ABI libraries load without a GPU, export every symbol that
include/sppark_amd.h declares, honour the by-value {int, char*} error protocol
and FAIL LOUDLY (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import have_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sppark_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(\w+)\s*\([^;{}]*\)\s*;", hdr)) - {"defined"})


def test_header_symbols_exported(libs):
    syms = declared_symbols()
    assert "mult_pippenger_inf" in syms and "compute_ntt" in syms and "cuda_available" in syms
    common = {"cuda_available", "drop_gpu_ptr_t", "clone_gpu_ptr_t", "drop_error_message", "cuda_func",
              "sppark_gpu_ptr_alloc", "sppark_gpu_ptr_get"}
    msm_only = {s for s in syms if "msm" in s or "pippenger" in s or s.startswith("sppark_g1") or s.startswith("sppark_g2")} | {"sppark_ngpus", "sppark_batch_addition"}
    ntt_only = {"compute_ntt", "sppark_ntt", "sppark_lde", "sppark_lde_powers", "sppark_lde_expand", "sppark_ntt_release_cached",
                "sppark_ntt_cached_scratch_bytes", "sppark_ntt_cached_tables",
                "sppark_prefix_op", "sppark_poly_evaluate", "sppark_div_by_x_minus_z"}
    assert set(syms) == common | msm_only | ntt_only
    poly_only = {"sppark_prefix_op", "sppark_poly_evaluate", "sppark_div_by_x_minus_z"}
    for name, path in libs.items():
        L = ctypes.CDLL(path)
        if name in ("m31", "bb31x4"):                           # field types without NTT parameters: polynomial primitives only
            for s in common | poly_only:
                assert hasattr(L, s), (name, s)
            assert not hasattr(L, "compute_ntt")
            continue
        want = common | ntt_only | ((msm_only - ({s for s in msm_only if "fp2" in s or "_g2_" in s} if name in ("pallas", "vesta") else set()))
                                   if name in ("bls12_381", "bn254", "bls12_377", "pallas", "vesta") else set())   # curve libs: NTT over Fr too
        for s in want:
            assert hasattr(L, s), (name, s)
        # the device test hooks live in separate test libraries (libsppark_*_devtest.so)
        for s in ("sppark_devtest_field_op", "sppark_devtest_small_field_op", "sppark_devtest_xyzz_op"):
            assert not hasattr(L, s), (name, s)


def test_symbols_resolve_like_go_dlsym(libs):
    """go/sppark.go:83-96 resolves every registered name with dlsym on a handle
    opened next to the executable."""
    h = ctypes.CDLL(libs["bls12_381"], mode=ctypes.RTLD_LOCAL)
    for s in ("mult_pippenger_inf", "mult_pippenger", "cuda_available", "drop_error_message"):
        assert ctypes.cast(getattr(h, s), ctypes.c_void_p).value


@pytest.mark.skipif(have_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly(libs):
    import sppark_amd
    from sppark_amd import ffi
    assert sppark_amd.cuda_available() is False
    pts = np.zeros((4, 104), dtype=np.uint8)
    sc = np.zeros((4, 32), dtype=np.uint8)
    with pytest.raises(ffi.SpparkError) as e:
        sppark_amd.multi_scalar_mult_arkworks(pts, sc)
    assert e.value.code != 0 and e.value.message              # negated HIP error + text
    with pytest.raises(ffi.SpparkError):
        sppark_amd.NTT(0, np.zeros(8, dtype=np.uint64), sppark_amd.NTTInputOutputOrder.NN)


def test_error_out_is_infinity_and_message_owned(libs):
    """on error `out` is set to infinity (msm/pippenger.cuh:740) and the message
    is a malloc'ed string the caller frees (drop_error_message)."""
    if have_gpu():
        pytest.skip("needs the no-device error path")
    from sppark_amd import ffi
    L = ffi.load("bls12_381")
    out = np.full(144, 0xff, dtype=np.uint8)
    err = L.mult_pippenger_inf(out.ctypes.data, 0, 4, 0, 104)
    assert err.code != 0
    assert (out == 0).all()
    assert ctypes.string_at(err.message)
    L.drop_error_message(err.message)


def test_rccl_exchange_rejects_a_missing_communicator(libs):
    """sppark_msm_rccl_sum / sppark_msm_rccl (the one-process-per-GPU exchange): without a communicator the call is an
    error (EINVAL) with `out` at infinity and an owned message -- it never touches RCCL, which the libraries do not link
    (bound at run time, csrc/util/rccl_dyn.hpp)."""
    import errno
    import subprocess
    from sppark_amd import ffi
    for name in ("bls12_381", "pallas"):
        L = ffi.load(name)
        fb = 48 if name == "bls12_381" else 32
        part = np.zeros(3 * fb, dtype=np.uint8)
        out = np.full(3 * fb, 0xff, dtype=np.uint8)
        err = L.sppark_msm_rccl_sum(out.ctypes.data, part.ctypes.data, 0, None, None)
        assert err.code == errno.EINVAL and (out == 0).all() and b"communicator" in ctypes.string_at(err.message)
        L.drop_error_message(err.message)
        needed = subprocess.run(["readelf", "-d", libs[name]], capture_output=True, text=True).stdout
        assert "rccl" not in needed.lower()
    L = ffi.load("pallas")                                     # a curve without G2
    out = np.full(96, 0xff, dtype=np.uint8)
    err = L.sppark_msm_rccl_sum(out.ctypes.data, out.ctypes.data, 1, None, None)
    assert err.code != 0 and (out == 0).all()
    L.drop_error_message(err.message)


def test_host_point_helpers(libs, oracle):
    """sppark_g1_jacobian_sum / sppark_g1_to_affine are host arithmetic (the
    multi-GPU combine step) and must agree with the oracle."""
    import sppark_amd
    O = oracle
    for curve, name in ((O.BLS12_381, "bls12_381"), (O.BN254, "bn254")):
        fb = O.FP_BYTES[curve]
        pts = O.g1_gen_points(curve, 6, 42)
        jac = np.zeros((6, 3 * fb), dtype=np.uint8)
        one = O.field_op(O.FIELD_BLS_FP if curve == 0 else O.FIELD_BN_FP, 4, O.int_to_limbs(1, fb)).view(np.uint8)
        jac[:, :2 * fb] = pts; jac[:, 2 * fb:] = one
        jac[4] = 0                                              # infinity
        jac[5] = jac[0]                                         # forces the doubling branch
        s = sppark_amd.jacobian_sum(jac, name)
        acc = np.zeros(3 * fb, dtype=np.uint8)
        for j in jac:
            acc = O.jac_add(curve, acc, j)
        assert O.jac_eq(curve, s, acc)
        assert (sppark_amd.to_affine(s, name) == O.jac_to_affine(curve, acc)).all()
        assert (sppark_amd.to_affine(np.zeros(3 * fb, dtype=np.uint8), name) == 0).all()


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/sppark_amd.h -- the drop-in boundary -- compiles as C99 and as C++ with warnings as errors: plain
    pointers and sizes, no torch / HIP types in any signature (what a cgo / bindgen / ctypes binding consumes)."""
    import shutil
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    body = '#include "sppark_amd.h"\nint main(void) { return sizeof(SppError) == 0; }\n'
    for cc, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "cpp")):
        if shutil.which(cc) is None:
            pytest.skip(cc + " not available")
        src = tmp_path / ("t." + ext)
        src.write_text(body)
        subprocess.check_call([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)])


def test_rccl_abi_constants_match_the_installed_header(tmp_path):
    """csrc/util/rccl_dyn.hpp declares the handful of NCCL/RCCL ABI facts it needs itself (so that the libraries build
    without the rccl development headers): hold them against <rccl/rccl.h> where that header is installed."""
    import subprocess
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl development header on this machine")
    src = tmp_path / "rccl_abi.cpp"
    src.write_text('''
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <type_traits>
static_assert((int)ncclSuccess == 0, "ncclSuccess");
static_assert((int)ncclUint8 == 1 && (int)ncclInt8 == 0, "ncclUint8");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int), "enums are ints");
static_assert(std::is_pointer<ncclComm_t>::value, "opaque communicator handle");
static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclCommCount), ncclResult_t (*)(const ncclComm_t, int*)>::value, "ncclCommCount");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString");
int main() { return 0; }
''')
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "--cuda-host-only", "-std=c++17", "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_g2_path_switch_validates_its_argument(libs):
    """sppark_msm_g2_path (which accumulation kernel mult_pippenger_fp2_inf runs; a test hook, process-wide): 0 / 1 / 2 are
    accepted, anything else is hipErrorInvalidValue with an owned message and leaves the setting alone; the curves
    without G2 do not export it.  (No device is touched.)"""
    from sppark_amd import ffi
    for name in ("bls12_381", "bn254", "bls12_377"):
        L = ffi.load(name)
        for mode in (1, 2, 0):
            err = L.sppark_msm_g2_path(mode)
            assert err.code == 0 and not err.message
        err = L.sppark_msm_g2_path(3)
        assert err.code != 0 and err.message
        L.drop_error_message(err.message)
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", libs["pallas"]], capture_output=True, text=True).stdout
    assert "sppark_msm_g2_path" not in syms and "mult_pippenger_fp2_inf" not in syms
