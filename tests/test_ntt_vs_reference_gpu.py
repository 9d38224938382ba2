"""GPU: sppark_amd's NTT against THE REFERENCE'S OWN NTT running on the same MI355X.

The reference ships a HIP path for its NTT (ff/gl64_t.hip, ff/mont32_t.hip, ff/mont_t.hip behind -include
util/cuda2hip.hpp; rust/src/build.rs:69-112).  oracle/Makefile (`ref_ntt`) compiles poc/ntt-cuda/cuda/ntt_api.cu and
util/all_gpus.cpp for gfx950 from the sources where they lie into oracle/_ref/libref_ntt_<field>.so (git-ignored test
infrastructure that travels to the GPU box prebuilt; /root/reference itself is not needed at run time).  NTT outputs are
unique bit patterns, so equality with the reference's output is exact: this pins the NTT half of the parity claim by the
reference itself, not by the oracle's restatement.  The MSM has no such path: msm/pippenger.cuh is CUDA-only
(PTX inline assembly), and its CPU twin needs blst for the host field.
"""
import numpy as np
import pytest

import recipe

pytestmark = pytest.mark.gpu
# (library of sppark_amd == library of the reference build, element kind)
LIBS = ["gl64", "gl64_plonky2", "bb31", "bb31_canonical", "bls12_381", "bn254", "bls12_377", "pallas", "vesta"]
KIND = {"gl64_plonky2": "gl64", "bb31_canonical": "bb31"}


def _need(O, lib):
    if not O.ref_ntt_available(lib):
        pytest.skip("oracle/_ref/libref_ntt_%s.so is not built (oracle/Makefile ref_ntt needs /root/reference at BUILD time)" % lib)


@pytest.mark.parametrize("lib", LIBS)
def test_ntt_all_modes_equal_the_reference_build(oracle, libs, lib):
    """compute_ntt on a host buffer, every order x direction x type (ntt/ntt.cuh:33-36), ours == the reference's"""
    import sppark_amd
    O = oracle
    _need(O, lib)
    kind = KIND.get(lib, lib)
    small = kind in ("gl64", "bb31")
    for lg in (list(range(1, 15)) + [16, 18, 20, 22]) if small else (list(range(1, 13)) + [14, 16]):
        x = recipe.ntt_input(kind, lg, 1000 + lg)
        for order in range(4):
            for direction in range(2):
                for typ in range(2):
                    if lg > 16 and (typ == 1 or direction == 1) and order != 1:
                        continue
                    y = x.copy()
                    sppark_amd.compute_ntt(0, y, order, direction, typ, lib)
                    ref = O.ref_compute_ntt(lib, x, order, direction, typ)
                    assert (y == ref).all(), (lib, lg, order, direction, typ)


@pytest.mark.parametrize("lib", ["gl64", "bb31", "bls12_381", "bn254"])
def test_ntt_full_size_equals_the_reference_build(oracle, libs, lib):
    """BASELINE configs[1] / [4] (2^24 elements; 2^22 for the 256-bit fields): the whole output array against the
    reference's, forward NR, inverse RN of it (== the input), forward NN, coset forward NN"""
    import sppark_amd
    O = oracle
    _need(O, lib)
    lg = 24 if lib in ("gl64", "bb31") else 22
    rng = np.random.default_rng(24)
    if lib == "gl64":
        x = (rng.integers(0, 1 << 63, size=1 << lg, dtype=np.uint64) * 2 + 1) % np.uint64(O.GL64_P)
    elif lib == "bb31":
        x = (rng.integers(0, 1 << 32, size=1 << lg, dtype=np.uint64) % O.BB31_P).astype(np.uint32)
    else:                                                           # limbs of values below 2^252 < r (Montgomery residues)
        x = rng.integers(0, 1 << 63, size=(1 << lg, 4), dtype=np.uint64)
        x[:, 3] >>= 11
    for order, direction, typ in ((1, 0, 0), (0, 0, 0), (0, 0, 1), (3, 1, 1)):
        y = x.copy()
        sppark_amd.compute_ntt(0, y, order, direction, typ, lib)
        ref = O.ref_compute_ntt(lib, x, order, direction, typ)
        assert (y == ref).all(), (lib, order, direction, typ)
        if order == 1:
            back = O.ref_compute_ntt(lib, y, 2, 1, 0)               # the reference inverts OUR forward output
            assert (back == x).all()


@pytest.mark.parametrize("lib", ["gl64", "bb31", "bn254"])
def test_ntt_on_device_memory_equals_the_reference_build(oracle, libs, lib):
    """the device-pointer forms: sppark_ntt against NTT::Base_dev_ptr (ntt/ntt.cuh:344-350), both on tensors of the
    same device -- no host copy of either library in the path"""
    import torch
    import sppark_amd
    O = oracle
    _need(O, lib)
    for lg in (9, 13, 17, 21):
        x = recipe.ntt_input(lib, lg, 31 + lg) if lg <= 13 or lib != "bn254" else None
        if x is None:
            rng = np.random.default_rng(lg)
            x = rng.integers(0, 1 << 63, size=(1 << lg, 4), dtype=np.uint64); x[:, 3] >>= 11
        host = np.ascontiguousarray(x).view(np.int32 if x.dtype == np.uint32 else np.int64).reshape(-1)
        for order, direction, typ in ((1, 0, 0), (2, 1, 0), (0, 0, 1), (3, 1, 0)):
            a = torch.from_numpy(host.copy()).cuda()
            b = torch.from_numpy(host.copy()).cuda()
            torch.cuda.synchronize()
            sppark_amd.compute_ntt(0, a, order, direction, typ, lib)
            O.ref_ntt_dev(lib, b.data_ptr(), lg, order, direction, typ)
            torch.cuda.synchronize()
            assert torch.equal(a, b), (lib, lg, order, direction, typ)


@pytest.mark.parametrize("lib,lg,lgb", [("gl64", 10, 2), ("gl64", 15, 1), ("gl64_plonky2", 11, 1), ("bb31", 12, 3), ("bb31_canonical", 10, 2),
                                        ("bls12_381", 9, 2), ("bn254", 11, 1), ("pallas", 10, 1)])
def test_lde_equals_the_reference_build(oracle, libs, lib, lg, lgb):
    """NTT::LDE_aux (ntt/ntt.cuh:280-342): the extended evaluations and the aux output (bit-reversed coefficients)"""
    import sppark_amd
    O = oracle
    _need(O, lib)
    kind = KIND.get(lib, lib)
    x = recipe.ntt_input(kind, lg, 77 + lg)
    ref, ref_aux = O.ref_lde(lib, x, lgb, want_aux=True)
    w = 4 if x.ndim == 2 else 1
    buf = np.zeros(((1 << (lg + lgb)), w), dtype=x.dtype); buf[:1 << lg] = x.reshape(-1, w)
    aux = np.zeros((1 << lg, w), dtype=x.dtype)
    sppark_amd.LDE(0, buf, lg, lgb, lib, aux_out=aux)
    assert (buf.reshape(-1) == ref.reshape(-1)).all()
    assert (aux.reshape(-1) == ref_aux.reshape(-1)).all()
    assert (O.ref_lde(lib, x, lgb).reshape(-1) == ref.reshape(-1)).all()
