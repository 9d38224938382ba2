"""CPU: NTT kernel phase functions run on the host (tests/emu/emu_ntt.cpp) vs the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import recipe

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _emu(feature):
    so = os.path.join(HERE, "emu", "libemu_ntt_%s.so" % feature)
    src = os.path.join(HERE, "emu", "emu_ntt.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "sppark_amd", "csrc")
    newest = max([os.stat(src).st_mtime] + [os.stat(os.path.join(r, f)).st_mtime for r, _, fs in os.walk(csrc) for f in fs])
    if not os.path.exists(so) or os.stat(so).st_mtime < newest:
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-DFEATURE_" + feature, "-o", so, src])
    L = ctypes.CDLL(so)
    L.emu_ntt.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
    L.emu_lde.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
    L.emu_ntt_plan.argtypes = [ctypes.c_uint, ctypes.c_uint]
    L.emu_ntt_plan.restype = None
    L.emu_ntt_lat.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    L.emu_ntt_lat.restype = None
    L.emu_ntt_lat_plan.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]
    L.emu_ntt_lat_plan.restype = ctypes.c_uint
    L.emu_ntt_small.argtypes = [ctypes.c_uint]
    L.emu_ntt_small.restype = None
    L.emu_ntt_small_sized.argtypes = [ctypes.c_uint]
    L.emu_ntt_small_sized.restype = None
    L.emu_ntt_coset_fold.argtypes = [ctypes.c_uint]
    L.emu_ntt_coset_fold.restype = None
    L.emu_ntt_coset_mode.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    L.emu_ntt_coset_mode.restype = ctypes.c_uint
    return L


FIELDS = [("gl64", "GOLDILOCKS"), ("bb31", "BABY_BEAR"), ("bls12_381", "BLS12_381"), ("bn254", "BN254")]


def _oracle_ntt(O, field):
    if field in ("bls12_381", "bn254"):
        curve = O.BLS12_381 if field == "bls12_381" else O.BN254
        return lambda x, order, direction, typ: O.ntt_fr(curve, x, order, direction, typ)
    return O.ntt_gl64 if field == "gl64" else O.ntt_bb31


@pytest.mark.parametrize("field,feature", FIELDS)
def test_small_transforms_in_one_work_group_on_host(oracle, field, feature):
    """k_ntt_small (what ntt_engine::run launches up to 2^10 elements: a butterfly pair per lane in registers, loaded with
    the order's permutation and the coset powers, a value swapped with the lane at distance 2^d after every stage,
    stored with 1/n, the inverse coset powers and RR's permutation), its lanes as host threads: every size 2^1 .. 2^11
    (256-bit fields: 2^10), every order, direction and type against the oracle, with the lane counts the driver uses."""
    L = _emu(feature)
    f = _oracle_ntt(oracle, field)
    cap = 10 if field in ("bls12_381", "bn254") else 11         # ntt_small_cap<F>: what the kernel is compiled for
    L.emu_ntt_small(cap)
    # single-word fields: the instances compiled for one size (2^8 ... 2^11, what the driver launches there) and the
    # run-time-size instance below them; then the run-time-size instance at those sizes too
    wide = field in ("bls12_381", "bn254")
    for sized in (1,) if wide else (1, 0):
        L.emu_ntt_small_sized(sized)
        for lg in range(1 if sized else 8, cap + 1):
            x = recipe.ntt_input(field, lg, 900 + lg)
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        y = x.copy()
                        L.emu_ntt(y.ctypes.data, lg, order, direction, typ, 64)
                        assert (y == f(x, order, direction, typ)).all(), (field, sized, lg, order, direction, typ)
            # forward then inverse is the identity in every order pairing the reference's tests use (poc/ntt-cuda/tests/ntt.rs)
            y = x.copy()
            L.emu_ntt(y.ctypes.data, lg, 1, 0, 1, 64); L.emu_ntt(y.ctypes.data, lg, 2, 1, 1, 64)
            assert (y == x).all()
    L.emu_ntt_small_sized(1)
    L.emu_ntt_small(9 if field in ("bls12_381", "bn254") else 11)  # the engine's defaults


@pytest.mark.parametrize("field,feature", [("gl64", "GOLDILOCKS"), ("bb31", "BABY_BEAR"),
                                           ("bls12_381", "BLS12_381"), ("bn254", "BN254")])
def test_ntt_kernels_on_host(oracle, field, feature):
    O = oracle
    L = _emu(feature)
    if field in ("bls12_381", "bn254"):
        curve = O.BLS12_381 if field == "bls12_381" else O.BN254
        f = lambda x, order, direction, typ: O.ntt_fr(curve, x, order, direction, typ)
    else:
        f = O.ntt_gl64 if field == "gl64" else O.ntt_bb31
    wide = field in ("bls12_381", "bn254")
    L.emu_ntt_small(0)                                          # the general path at every size (k_ntt_small has its own test)
    # the 256-bit fields: the engine's default passes (one stage per round, up to 8 stages) and the register passes
    # (radix-4 x radix-4; SPPARK_NTT_LAT_SMAX=0)
    for lat in ((8, 0) if wide else (0,)):
        L.emu_ntt_lat(lat, -1, -1)
        for lg in (list(range(1, 15)) + [16]) if not wide else (list(range(1, 12)) + [13]):
            x = recipe.ntt_input(field, lg, 100 + lg)
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        # (above 2^13: standard transforms only; NN / RR -- the tiled bit reversal with several `mid`
                        # bits, in its 16-byte form for the single-word fields -- in the forward direction only)
                        if lg > 13 and (typ == 1 or (order in (0, 3) and direction == 1)):
                            continue
                        y = x.copy()
                        L.emu_ntt(y.ctypes.data, lg, order, direction, typ, 64 if lg < 10 else 256)
                        assert (y == f(x, order, direction, typ)).all(), (field, lat, lg, order, direction, typ)
    L.emu_ntt_lat(8 if wide else 0, -1, -1)
    L.emu_ntt_small(9 if wide else 11)                          # the engine's defaults


@pytest.mark.parametrize("field,feature", [("gl64", "GOLDILOCKS"), ("bb31", "BABY_BEAR"),
                                           ("bls12_381", "BLS12_381"), ("bn254", "BN254")])
def test_lde_kernels_on_host(oracle, field, feature):
    """ntt_engine::lde()'s kernel sequence (inverse NR passes, lde_spread_item, forward RN
    passes) executed on the host against the oracle's restatement of NTT::LDE_aux."""
    O = oracle
    L = _emu(feature)
    w = 4 if field in ("bls12_381", "bn254") else 1
    # (single-word fields from 2^12 extended elements on, blow-up 2 / 4 / 8: the spread and the coset shift inside the first
    #  k_ntt12 step of the forward transform -- emu_lde returns 1 --, a spread pass of its own otherwise)
    for lg, lgb in ((1, 1), (3, 2), (6, 1), (9, 2), (11, 3), (11, 1), (10, 2), (12, 2), (9, 4), (13, 1)) if w == 1 else ((1, 1), (3, 2), (6, 1), (9, 2), (8, 3)):
        x = recipe.ntt_input(field, lg, 300 + lg)
        exp, aux_exp = O.lde(field, x, lgb, want_aux=True)
        buf = np.zeros(((1 << (lg + lgb)), w), dtype=x.dtype); buf[:1 << lg] = x.reshape(-1, w)
        aux = np.zeros((1 << lg, w), dtype=x.dtype)
        fused = L.emu_lde(buf.ctypes.data, lg, lgb, aux.ctypes.data)
        assert fused == (1 if w == 1 and lg + lgb >= 12 and lgb <= 3 else 0), (field, lg, lgb, fused)
        assert (buf.reshape(exp.shape) == exp).all(), (field, lg, lgb)
        assert (aux.reshape(aux_exp.shape) == aux_exp).all(), (field, lg, lgb)


@pytest.mark.parametrize("field,feature", [("gl64", "GOLDILOCKS"), ("bb31", "BABY_BEAR")])
def test_ntt_radix64_plan_on_host(oracle, field, feature):
    """The radix-64 plan (k_ntt6 / k_ntt12 round functions, r64_table_item tables, make_r64_plan) on the
    host against the oracle AND against the 8-stage plan: 12-stage kernel alone (2^12), a generic pass
    above it (2^13..2^17), k_ntt6 with ONE inter-pass table (2^18, direct) and with the two small
    tables (2^18, factored), both directions and all four orders."""
    O = oracle
    L = _emu(feature)
    f = O.ntt_gl64 if field == "gl64" else O.ntt_bb31
    try:
        for lg, direct, modes in ((12, 20, "all"), (13, 20, "all"), (15, 20, "nr"), (17, 20, "nr"),
                                  (18, 20, "all"), (18, 12, "nr"), (19, 12, "nr")):
            x = recipe.ntt_input(field, lg, 500 + lg)
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        if modes == "nr" and (typ == 1 or order in (0, 3)):
                            continue
                        if lg >= 18 and modes == "all" and typ == 1 and order != 1:
                            continue
                        L.emu_ntt_plan(12, direct)
                        y = x.copy()
                        L.emu_ntt(y.ctypes.data, lg, order, direction, typ, 256)
                        assert (y == f(x, order, direction, typ)).all(), (field, lg, direct, order, direction, typ)
            # the 8-stage plan gives the same bytes
            L.emu_ntt_plan(99, 20)
            z = x.copy()
            L.emu_ntt(z.ctypes.data, lg, 1, 0, 0, 256)
            assert (z == f(x, 1, 0, 0)).all()
    finally:
        L.emu_ntt_plan(12, 20)


@pytest.mark.parametrize("field,feature", [("gl64", "GOLDILOCKS"), ("bb31", "BABY_BEAR")])
def test_ntt_coset_folded_into_radix64_passes_on_host(oracle, field, feature):
    """Coset transforms on a plan of k_ntt6 / k_ntt12 steps carry the powers of the coset generator in their twiddle tables
    and 64 constants instead of a separate scaling pass (r64_coset_mode / r64_table_item / r64_cz_item): the four foldable
    cases -- forward DIF (NR) and inverse DIT (RN, NN) with natural exponents, inverse DIF (NR) and forward DIT (RN, NN)
    with bit-reversed ones -- at 2^12 (k_ntt12 alone: bit-reversed exponents only), 2^18 (k_ntt6 + k_ntt12) with the
    step's twiddles from one table and from the two small ones, and with a generic pass on top of the plan (2^13: one
    stage in a single round, 2^14: two stages, 2^15 / 2^17: three / five, table and generated twiddles; ntt_pass::cmode),
    against the oracle; RR keeps the separate pass; with the fold switched off the same bytes."""
    O = oracle
    L = _emu(feature)
    f = O.ntt_gl64 if field == "gl64" else O.ntt_bb31
    try:
        for lg, direct in ((12, 20), (18, 20), (18, 12), (13, 20), (14, 20), (15, 20), (17, 20)):
            x = recipe.ntt_input(field, lg, 700 + lg)
            L.emu_ntt_plan(12, direct)
            for order in range(4):
                for direction in range(2):
                    exp = f(x, order, direction, 1)
                    # NR forward / RN, NN inverse: natural exponents (1); NR inverse / RN, NN forward: bit-reversed (2)
                    want = 0 if order == 3 else 1 if (order == 1) != (direction == 1) else 2
                    if lg == 12 and want == 1:
                        want = 0
                    L.emu_ntt_coset_fold(1)
                    assert L.emu_ntt_coset_mode(lg, order, direction) == want, (lg, order, direction)
                    for fold in ((1, 0) if (lg, direct) in ((12, 20), (18, 20)) and order == 1 else (1,)):
                        L.emu_ntt_coset_fold(fold)
                        y = x.copy()
                        L.emu_ntt(y.ctypes.data, lg, order, direction, 1, 256)
                        assert (y == exp).all(), (field, lg, direct, order, direction, fold)
    finally:
        L.emu_ntt_plan(12, 20); L.emu_ntt_coset_fold(1)


@pytest.mark.parametrize("field,feature", [("bls12_381", "BLS12_381"), ("bn254", "BN254"), ("bb31", "BABY_BEAR"), ("gl64", "GOLDILOCKS")])
def test_ntt_one_stage_per_round_passes_on_host(oracle, field, feature):
    """k_ntt_pass_lat's rounds (load, S stages in LDS with one butterfly per lane, store) under several pass shapes:
    stages per pass 8 / 6 / 5 / 3, tile rows of 1 / 4 / 16 columns, tiles that hold several sub-problems -- every order,
    direction and type against the oracle.  (The kernel is launched for the 256-bit fields only; BabyBear and
    Goldilocks run the same index math here -- Goldilocks with its sign-folded power-of-two roots, root_neg, which exist
    up to order 64: passes of at most 6 stages.)"""
    O = oracle
    L = _emu(feature)
    if field in ("bls12_381", "bn254"):
        curve = O.BLS12_381 if field == "bls12_381" else O.BN254
        f = lambda x, order, direction, typ: O.ntt_fr(curve, x, order, direction, typ)
    else:
        f = O.ntt_gl64 if field == "gl64" else O.ntt_bb31
    try:
        L.emu_ntt_small(0)                                          # the passes, also where the engine runs k_ntt_small
        L.emu_ntt_plan(99, 20)                                      # (Goldilocks: generic passes, not the radix-64 plan)
        for smax, lgc, lgtile in ((8, -1, -1), (8, 2, 10), (8, 0, 8), (6, 4, 11), (5, 1, 9), (3, 2, 6)):
            if field == "gl64" and smax > 6:
                continue
            L.emu_ntt_lat(smax, lgc, lgtile)
            for lg in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13):
                if lg > 11 and (smax, lgc) not in ((8, -1), (6, 4)):
                    continue
                x = recipe.ntt_input(field, lg, 500 + lg)
                for order in range(4):
                    for direction in range(2):
                        for typ in range(2):
                            if lg > 10 and (typ == 1 or order in (0, 3)):
                                continue
                            y = x.copy()
                            L.emu_ntt(y.ctypes.data, lg, order, direction, typ, 64 if lg < 9 else 256)
                            assert (y == f(x, order, direction, typ)).all(), (field, smax, lgc, lgtile, lg, order, direction, typ)
    finally:
        L.emu_ntt_lat(8 if field in ("bls12_381", "bn254") else 0, -1, -1)      # the engine's defaults
        L.emu_ntt_plan(12, 20)
        L.emu_ntt_small(9 if field in ("bls12_381", "bn254") else 11)


def test_one_stage_per_round_plan_invariants():
    """make_ntt_lat_plan (what ntt_engine::run launches k_ntt_pass_lat with) for every size a 256-bit field can have and
    the shapes the knobs allow: the passes' stages add up to lg n, each splits what the previous one left, a tile never
    exceeds 2048 elements (1024 lanes at one butterfly per lane, 64 KB of LDS at 32 bytes per element), its row is inside
    the sub-problem's row (or the tile holds whole sub-problems), and the default shape gives a transform of 2^16 or more
    elements at least 256 tiles."""
    L = _emu("BLS12_381")
    out = (ctypes.c_uint * 64)()
    for smax, lgc, lgtile in ((8, -1, -1), (6, -1, -1), (4, -1, -1), (8, 2, 10), (8, 0, 8), (8, 3, 11), (7, 4, 11), (5, 4, 9), (1, -1, -1)):
        for lg in range(1, 33):
            if (lg + smax - 1) // smax > 16:
                continue                                            # (more passes than a plan holds: smax = 1 above 2^16)
            npass = L.emu_ntt_lat_plan(lg, smax, lgc, lgtile, out)
            assert 1 <= npass <= 16
            rem = lg
            for i in range(npass):
                lg_cur, S, lgC, lgG = out[4 * i:4 * i + 4]
                assert lg_cur == rem and 1 <= S <= smax and S <= rem, (smax, lg, i)
                lgQ = lg_cur - S
                assert lgC <= lgQ and (lgG == 0 or lgC == lgQ), (smax, lg, i)
                assert lgG <= lg - lg_cur, (smax, lg, i)            # whole sub-problems that exist
                tile = lgG + S + lgC
                assert tile <= 11 and tile <= lg, (smax, lgc, lgtile, lg, i, tile)
                if lgc < 0 and lg >= 16 and i == 0:
                    assert lg - tile >= 8, (smax, lg, tile)         # >= 256 tiles
                rem -= S
            assert rem == 0
            assert max(out[4 * i + 1] for i in range(npass)) - min(out[4 * i + 1] for i in range(npass)) <= 1      # near-equal split
