"""CPU: run the product's per-work-item kernel bodies on the host
(tests/emu/emu_msm.cpp, built with -DSPPARK_HOST_EMULATION) and compare with the
oracle.  This checks the device field/point arithmetic at C++ level, the digit
recoding, the fixed-run accumulate / record-tree logic and the bucket-sum levels
in the GPU-less container; the real kernels are checked on the GPU by
tests/test_msm_gpu.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import recipe

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _emu(feature, g2=False):
    so = os.path.join(EMU, "libemu_%s%s.so" % (feature, "_G2" if g2 else ""))
    src = os.path.join(EMU, "emu_msm.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "sppark_amd", "csrc")
    newest = max(os.stat(os.path.join(r, f)).st_mtime for r, _, fs in os.walk(csrc) for f in fs)
    newest = max(newest, os.stat(src).st_mtime)
    if not os.path.exists(so) or os.stat(so).st_mtime < newest:
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared",
                               "-DFEATURE_" + feature] + (["-DSPPARK_G2"] if g2 else []) + ["-o", so, src])
    L = ctypes.CDLL(so)
    vp, sz, ci, cu = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint
    L.emu_field_op.argtypes = [ci, ci, vp, vp, vp, sz]
    L.emu_xyzz_op.argtypes = [ci, vp, vp, vp, sz]
    L.emu_msm.argtypes = [vp, vp, sz, sz, vp, ci, cu, cu, cu, cu, cu, ci, vp, cu]
    L.emu_pairs_check.argtypes = [vp, sz, sz]
    L.emu_fixed_base.argtypes = [vp, vp, sz, sz, vp, cu]
    L.emu_fp2x_op.argtypes = [ci, ci, vp, vp, vp, sz]
    L.emu_g2_chain.argtypes = [vp, vp, sz, sz]
    L.emu_g2c_chain.argtypes = [vp, vp, vp, sz, sz, cu]
    L.emu_g2c.argtypes = [ci]
    L.emu_g2c.restype = None
    return L


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("curve", [0, 1])
def test_device_field_code_on_host(oracle, curve):
    O = oracle
    L = _emu("BLS12_381" if curve == 0 else "BN254")
    rng = np.random.default_rng(curve)
    for which, p, nb, ofield in ((0, O.FP_MODULUS[curve], O.FP_BYTES[curve], O.FIELD_BLS_FP if curve == 0 else O.FIELD_BN_FP),
                                 (1, O.FR_MODULUS[curve], 32, O.FIELD_BLS_FR if curve == 0 else O.FIELD_BN_FR)):
        n = 96
        va = [int.from_bytes(rng.bytes(nb + 8), "little") % p for _ in range(n)]
        vb = [int.from_bytes(rng.bytes(nb + 8), "little") % p for _ in range(n)]
        va[:5] = [0, 1, p - 1, p - 1, 0]; vb[:5] = [0, p - 1, p - 1, 1, 5]
        a = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in va), dtype=np.uint8).copy()
        b = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in vb), dtype=np.uint8).copy()
        for op, oop in ((0, 0), (1, 1), (2, 2), (3, 7), (4, 6), (5, 5), (6, 4)):
            out = np.zeros_like(a)
            L.emu_field_op(which, op, P(out), P(a), P(b), n)
            for i in range(n):
                e = O.field_op(ofield, oop, a[i * nb:(i + 1) * nb].view(np.uint64), b[i * nb:(i + 1) * nb].view(np.uint64))
                assert (e.view(np.uint8) == out[i * nb:(i + 1) * nb]).all(), (curve, which, op, i)


@pytest.mark.parametrize("curve,feature,which", [(0, "BLS12_381", 0), (1, "BN254", 0), (1, "BN254", 1)])
def test_wire_to_bucket_field_conversion_on_host(oracle, curve, feature, which):
    """montx_dev::from_std: wire words -> the bucket field's own limbs as a SHIFT by the domain offset and the subtraction of
    q p with an estimated quotient (ff/montx_dev.hpp; a product until round 5).  For every input the limbs must be normalised
    and their value congruent to w 2^SH and below 2p -- checked with big integers on the inputs where the estimate is at
    its edges: w 2^SH just below / at / above every multiple k p that the shift can reach (sampled for the wide shifts),
    0, 1, p - 1, powers of two, all-ones patterns below p, random values.  which = 1: the ten 28-bit limbs under alt_bn128's
    G2 pipeline (a domain offset of 24 bits: the widest quotient)."""
    O = oracle
    L = _emu(feature)
    p = O.FP_MODULUS[curve]
    nb = O.FP_BYTES[curve]
    info = np.zeros(3, dtype=np.uint32)
    probe = np.zeros(nb, dtype=np.uint8); out = np.zeros(64, dtype=np.uint32)
    assert L.emu_from_std_limbs(which, P(out), P(probe), 1, P(info)) == 1
    NL, LB, SH = (int(v) for v in info)
    assert (NL, LB) == ((14, 28) if curve == 0 else (10, 28) if which else (9, 29)) and SH == NL * LB - 8 * nb
    rng = np.random.default_rng(77 + curve)
    vals = {0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2}
    vals |= {1 << k for k in range(0, p.bit_length() - 1, 7)} | {(1 << k) - 1 for k in range(1, p.bit_length(), 5)}
    ks = range(1, (1 << SH) + 1) if SH <= 10 else [int(v) for v in rng.integers(1, 1 << SH, 600)] + [1, 2, (1 << SH) - 1, 1 << SH]
    for k in ks:                                                # w 2^SH around k p
        w0 = (k * p) >> SH
        vals |= {w for w in (w0 - 1, w0, w0 + 1, w0 + 2) if 0 <= w < p}
    vals |= {int.from_bytes(rng.bytes(nb + 8), "little") % p for _ in range(500)}
    vals = sorted(vals)
    a = np.frombuffer(b"".join(v.to_bytes(nb, "little") for v in vals), dtype=np.uint8).copy()
    out = np.zeros(len(vals) * NL, dtype=np.uint32)
    L.emu_from_std_limbs(which, P(out), P(a), len(vals), P(info))
    for i, w in enumerate(vals):
        limbs = [int(v) for v in out[i * NL:(i + 1) * NL]]
        assert all(l < (1 << LB) for l in limbs), (hex(w), limbs)
        v = sum(l << (LB * j) for j, l in enumerate(limbs))
        assert v < 2 * p and (v - (w << SH)) % p == 0, (hex(w), hex(v))


@pytest.mark.parametrize("curve", [0, 1])
def test_msm_pipeline_on_host(oracle, curve):
    O = oracle
    L = _emu("BLS12_381" if curve == 0 else "BN254")
    fb = O.FP_BYTES[curve]
    # (n, wbits, L, F, K, nslabs, flagged): automatic plan and adversarial tunables
    for n, wb, LL, F, K, ns, flagged in ((1, 0, 0, 0, 0, 0, False), (2, 0, 0, 0, 0, 0, True), (33, 0, 0, 0, 0, 0, False),
                                         (1000, 0, 0, 0, 0, 0, True), (1000, 7, 4, 4, 2, 3, False),
                                         (700, 9, 16, 8, 4, 2, True), (2048, 10, 8, 32, 8, 1, False),
                                         (300, 2, 4, 4, 2, 1, False), (300, 16, 64, 32, 8, 1, False), (500, 19, 8, 8, 8, 2, True)):
        pts, sc = recipe.msm_inputs(curve, n, 1234 + n + wb, flagged=flagged)
        out = np.zeros(3 * fb, dtype=np.uint8)
        L.emu_msm(P(out), P(pts), pts.shape[1], n, P(sc), 0, wb, LL, F, K, ns, 1, None, 0)
        assert (O.jac_to_affine(curve, out) == O.msm_affine(curve, pts, sc, algo=0, param=4)).all(), (n, wb, LL, F, K, ns)


def test_msm_pipeline_with_packed_sort_records_on_host(oracle):
    """Level-A sort records of 4 bytes (sign | index mod 2^IB | k_lo; csrc/msm/msm_sort_records.hpp): the index bits above IB
    are not stored, level B finds them from the POSITION of a record in its partition (the records lie in slab order; a
    binary search over the index groups' first positions).  The pipeline packs and unpacks with the product's own code, with
    an index field so short that a few hundred points spread over up to 8 index groups (every slab, resp. every pair of
    slabs, is one): skewed scalars leave groups EMPTY in a partition (equal boundaries), a ragged size a short last slab."""
    O = oracle
    curve = 1
    L = _emu("BN254")
    L.emu_msm_pack.argtypes = [ctypes.c_uint]; L.emu_msm_pack.restype = None
    fb = O.FP_BYTES[curve]
    try:
        for mode in (1, 2):
            L.emu_msm_pack(mode)
            for n, wb, LL, F, K, ns in ((1, 0, 0, 0, 0, 0), (33, 0, 0, 0, 0, 3), (1000, 7, 4, 4, 2, 7), (700, 9, 16, 8, 4, 5),
                                        (2048, 10, 8, 32, 8, 8), (300, 2, 4, 4, 2, 4), (500, 19, 8, 8, 8, 6)):
                pts, sc = recipe.msm_inputs(curve, n, 99 + n + wb, flagged=True)
                out = np.zeros(3 * fb, dtype=np.uint8)
                assert L.emu_msm(P(out), P(pts), pts.shape[1], n, P(sc), 0, wb, LL, F, K, ns, 1, None, 0) == 0
                assert (O.jac_to_affine(curve, out) == O.msm_affine(curve, pts, sc, algo=0, param=4)).all(), (mode, n, wb, ns)
            # all scalars equal: one bucket per window holds everything; and half of the scalars zero
            pts, sc = recipe.msm_inputs(curve, 600, 7, edge=False, flagged=True)
            s_eq = sc.copy(); s_eq[:] = sc[0]
            s_half = sc.copy(); s_half[::2] = 0
            for s_ in (s_eq, s_half):
                out = np.zeros(3 * fb, dtype=np.uint8)
                assert L.emu_msm(P(out), P(pts), pts.shape[1], 600, P(s_), 0, 8, 8, 4, 4, 6, 1, None, 0) == 0
                assert (O.jac_to_affine(curve, out) == O.msm_affine(curve, pts, s_, algo=0, param=4)).all(), mode
    finally:
        L.emu_msm_pack(0)


@pytest.mark.parametrize("curve", [2, 3, 5])
def test_msm_g2_pipeline_on_host(oracle, curve):
    """the same kernel bodies instantiated over Fp2 (ff/fp2x_dev.hpp + ec/xyzzx2_dev.hpp: the loosely-reduced
    28-bit-limb base field; u^2 = -1 for BLS12-381 / alt_bn128, u^2 = -5 for BLS12-377): G2 MSM on the host"""
    O = oracle
    L = _emu({2: "BLS12_381", 3: "BN254", 5: "BLS12_377"}[curve], g2=True)
    fb = O.FP_BYTES[curve]
    for n, wb, LL, F, K, ns, flagged in ((1, 0, 0, 0, 0, 0, True), (33, 0, 0, 0, 0, 0, False), (600, 0, 0, 0, 0, 0, True),
                                         (500, 7, 4, 4, 2, 3, False), (300, 11, 16, 8, 4, 2, True)):
        pts, sc = recipe.msm_inputs(curve, n, 4321 + n + wb, flagged=flagged)
        out = np.zeros(3 * fb, dtype=np.uint8)
        L.emu_msm(P(out), P(pts), pts.shape[1], n, P(sc), 0, wb, LL, F, K, ns, 1, None, 0)
        assert (O.jac_to_affine(curve, out) == O.msm_affine(curve, pts, sc, algo=0, param=4)).all(), (n, wb, LL, F, K, ns)
    # all points equal (doubling branch) and all scalars equal
    pts, sc = recipe.msm_inputs(curve, 400, 5, edge=False, flagged=True)
    same = pts.copy(); same[:] = pts[0]
    s_eq = sc.copy(); s_eq[:] = sc[0]
    for p_, s_ in ((same, sc), (pts, s_eq)):
        out = np.zeros(3 * fb, dtype=np.uint8)
        L.emu_msm(P(out), P(p_), p_.shape[1], 400, P(s_), 0, 8, 8, 4, 4, 2, 1, None, 0)
        assert (O.jac_to_affine(curve, out) == O.msm_affine(curve, p_, s_, algo=0, param=4)).all()


@pytest.mark.parametrize("curve", [2, 3, 5])
def test_g2_one_component_per_wave_on_host(oracle, curve):
    """ec/xyzz2_coop.hpp + msm/msm_g2c_kernels.hpp (NOT the default path: msm_tunables::g2_coop): the G2 bucket with one
    Fp2 component per wave, a pair of waves = 128 host threads meeting at the barriers of the exchange.
    (i) chains of set / madd steps on 64 lanes -- a lane that meets its start point again (the cooperative doubling), one
    that meets its negative (infinity, then a fresh start), points at infinity in the list -- leave BIT-IDENTICAL images
    to the serial class after every step; (ii) the whole G2 MSM on the host with the accumulation run by wave pairs
    equals the oracle, ragged chunk counts, flagged and plain records, the all-equal-points case included."""
    O = oracle
    L = _emu({2: "BLS12_381", 3: "BN254", 5: "BLS12_377"}[curve], g2=True)
    fb = O.FP_BYTES[curve]
    words = {2: 28, 3: 20, 5: 28}[curve] * 4                    # internal XYZZ image: 4 coordinates x 2 components x NL limbs
    pts, _sc = recipe.msm_inputs(curve, 70, 99, flagged=True)     # (edge cases on: the list holds points at infinity)
    steps = 6
    out = np.zeros((steps * 64, words), dtype=np.uint32); ref = np.zeros_like(out)
    assert L.emu_g2c_chain(P(out), P(ref), P(pts), pts.shape[1], pts.shape[0], steps) == steps
    assert (out == ref).all(), np.argwhere((out != ref).any(axis=1))[:8].ravel()
    assert ref.any()
    try:
        L.emu_g2c(1)
        # (every 64 chunks of a window are 128 host threads here: sizes kept small)
        for n, wb, LL, F, K, ns, flagged in ((1, 16, 4, 4, 2, 1, True), (70, 13, 4, 4, 2, 3, False), (300, 13, 16, 8, 4, 2, True)):
            p_, s_ = recipe.msm_inputs(curve, n, 4321 + n + wb, flagged=flagged)
            res = np.zeros(3 * fb, dtype=np.uint8)
            L.emu_msm(P(res), P(p_), p_.shape[1], n, P(s_), 0, wb, LL, F, K, ns, 1, None, 0)
            assert (O.jac_to_affine(curve, res) == O.msm_affine(curve, p_, s_, algo=0, param=4)).all(), (n, wb, LL, F, K, ns)
        p_, s_ = recipe.msm_inputs(curve, 150, 5, edge=False, flagged=True)
        same = p_.copy(); same[:] = p_[0]
        res = np.zeros(3 * fb, dtype=np.uint8)
        L.emu_msm(P(res), P(same), same.shape[1], 150, P(s_), 0, 13, 8, 4, 4, 2, 1, None, 0)
        assert (O.jac_to_affine(curve, res) == O.msm_affine(curve, same, s_, algo=0, param=4)).all()
    finally:
        L.emu_g2c(0)


def test_msm_skewed_scalars_on_host(oracle):
    """all-equal scalars / tiny scalars / 50% zeros: one bucket holds everything,
    the record tree must still reduce it (SURVEY 8(d) skew cases)."""
    O = oracle
    L = _emu("BLS12_381")
    n = 1500
    pts, sc = recipe.msm_inputs(0, n, 77, edge=False)
    cases = []
    s_eq = sc.copy(); s_eq[:] = sc[0]; cases.append(s_eq)
    s_small = np.zeros_like(sc); s_small[:, 0] = sc[:, 0] & 3; cases.append(s_small)
    s_half = sc.copy(); s_half[::2] = 0; cases.append(s_half)
    same_pts = pts.copy(); same_pts[:] = pts[0]
    for s in cases:
        for p in (pts, same_pts):
            out = np.zeros(144, dtype=np.uint8)
            L.emu_msm(P(out), P(p), 96, n, P(s), 0, 8, 8, 4, 4, 2, 1, None, 0)
            assert (O.jac_to_affine(0, out) == O.msm_affine(0, p, s, algo=0, param=4)).all()


def test_msm_montgomery_scalars_on_host(oracle):
    O = oracle
    L = _emu("BN254")
    pts, sc = recipe.msm_inputs(1, 200, 5)
    mont = np.zeros_like(sc)
    for i in range(200):
        mont[i] = O.field_op(O.FIELD_BN_FR, 4, sc[i].view(np.uint64)).view(np.uint8)
    out = np.zeros(96, dtype=np.uint8)
    L.emu_msm(P(out), P(pts), 64, 200, P(mont), 1, 0, 0, 0, 0, 0, 1, None, 0)
    assert (O.jac_to_affine(1, out) == O.msm_affine(1, pts, sc)).all()


def test_msm_short_segment_join_on_host(oracle):
    """k_join_runs (join_runs_item): with uniform scalars every bucket that straddles a chunk boundary is
    a segment of a few records -- the join resolves all of them and the fan-in tree has nothing left;
    with one bucket holding everything the segment is longer than the walk and goes through the tree.
    Same group element with and without the join in both cases."""
    O = oracle
    L = _emu("BLS12_381")
    n = 3000
    pts, sc = recipe.msm_inputs(0, n, 99, edge=False)
    s_eq = sc.copy(); s_eq[:] = sc[0]
    s_two = sc.copy(); s_two[: n // 2] = sc[0]; s_two[n // 2:] = sc[1]
    for s, wb, LL, expect_long in ((sc, 8, 16, False), (sc, 6, 8, None), (sc, 4, 4, None), (s_eq, 8, 8, True), (s_two, 8, 16, True)):
        exp = O.msm_affine(0, pts, s, algo=0, param=4)
        for join in (1, 0):
            out = np.zeros(144, dtype=np.uint8)
            stats = np.zeros(2, dtype=np.uint32)
            L.emu_msm(P(out), P(pts), 96, n, P(s), 0, wb, LL, 4, 4, 2, join, P(stats), 0)
            assert (O.jac_to_affine(0, out) == exp).all(), (wb, LL, join)
            if join and expect_long is not None:
                assert bool(stats[0]) == expect_long, (wb, LL, stats)
                if not expect_long:
                    assert stats[1] == 0            # no record left for the tree


@pytest.mark.parametrize("curve,feature", [(0, "BLS12_381"), (1, "BN254")])
def test_msm_small_window_bucket_sums_on_host(oracle, curve, feature):
    """The bucket sums of windows of up to 256 buckets straight from the buckets (k_bucket_small_bits_coop: the buckets of
    S_b = the 1-based numbers with bit b set, small_sums_member / small_sums_gather; empty buckets from the sort's offsets;
    b doublings per part) against the oracle: 2 .. 256 buckets per window, uniform and sparse scalars (most buckets empty),
    all scalars equal."""
    O = oracle
    L = _emu(feature)
    fb = O.FP_BYTES[curve]
    n = 700
    pts, sc = recipe.msm_inputs(curve, n, 555, edge=True)
    s_sparse = np.zeros_like(sc); s_sparse[::7] = sc[::7]
    s_eq = sc.copy(); s_eq[:] = sc[4]
    for s in (sc, s_sparse, s_eq):
        exp = O.msm_affine(curve, pts, s, algo=0, param=4)
        for wb in (2, 3, 4, 5, 8, 9):
            out = np.zeros(3 * fb, dtype=np.uint8)
            L.emu_msm(P(out), P(pts), pts.shape[1], n, P(s), 0, wb, 8, 4, 4, 2, 1, None, 2)
            assert (O.jac_to_affine(curve, out) == exp).all(), (curve, wb)


def test_msm_piece_tree_on_host(oracle):
    """The record list of a small MSM by the piece tree (msm_piece_kernels.hpp piece_level_item): few buckets of many
    entries, every bucket cut into tens of fixed-length runs, its pieces found from the sort's offsets and added up
    level by level.  Uniform scalars: no record is left for the fan-in tree, whatever the window / run length (aligned
    and unaligned bucket starts, buckets inside one run, runs that end with the list, odd piece counts); a bucket with
    more pieces than the tree was sized for (skewed scalars; a forced small limit) keeps its records and goes through
    the fan-in tree; zero digits (shorter lists); G1 results against the oracle in every case."""
    O = oracle
    L = _emu("BLS12_381")
    L.emu_msm_piece_cmax.argtypes = [ctypes.c_uint]; L.emu_msm_piece_cmax.restype = None
    L.emu_msm_piece_fuse.argtypes = [ctypes.c_size_t]; L.emu_msm_piece_fuse.restype = None
    n = 2000
    pts, sc = recipe.msm_inputs(0, n, 4321, edge=True)
    s_eq = sc.copy(); s_eq[:] = sc[0]
    s_mix = sc.copy(); s_mix[n // 3:] = sc[1]
    s_half = sc.copy(); s_half[::2] = 0
    s_small = np.zeros_like(sc); s_small[:, :3] = sc[:, :3]
    try:
        for s, what in ((sc, "uniform"), (s_eq, "equal"), (s_mix, "mix"), (s_half, "half zero"), (s_small, "24-bit")):
            exp = O.msm_affine(0, pts, s, algo=0, param=4)
            for wb, LL in ((4, 8), (5, 7), (6, 16), (3, 5), (8, 4), (7, 1)):
                for cmax in ((0, 2, 8, 4096) if what in ("uniform", "equal") else (0, 8)):
                    # fuse: the narrow end of the tree in k_piece_tail_coop's order (a work-group's buckets through every
                    # level before the next work-group): every level, from a middle level, never
                    for fuse in ((0, 1 << 30, 4096) if cmax in (0, 8) else (0, 1 << 30)):
                        L.emu_msm_piece_cmax(cmax)
                        L.emu_msm_piece_fuse(fuse)
                        out = np.zeros(144, dtype=np.uint8)
                        stats = np.zeros(2, dtype=np.uint32)
                        L.emu_msm(P(out), P(pts), 96, n, P(s), 0, wb, LL, 4, 4, 2, 2, P(stats), 0)
                        assert (O.jac_to_affine(0, out) == exp).all(), (what, wb, LL, cmax, fuse)
                        if what == "uniform" and cmax in (0, 4096):
                            assert stats[0] == 0 and stats[1] == 0, (wb, LL, cmax, stats)      # nothing left for the fan-in tree
                        if what == "equal" and cmax == 2 and LL <= 16:
                            assert stats[0] == 1 and stats[1] > 0
    finally:
        L.emu_msm_piece_cmax(0)
        L.emu_msm_piece_fuse(0)


@pytest.mark.parametrize("curve,feature", [(0, "BLS12_381"), (1, "BN254")])
def test_low_latency_point_ops_on_host(oracle, curve, feature):
    """add_pairs / dbl_pairs (products in interleaved pairs, unmasked quotient digits: the forms the top of the
    bucket sums uses) give the same canonical XYZZ values as add / dbl, incl. the doubling and P + (-P) branches"""
    L = _emu(feature)
    pts, _ = recipe.msm_inputs(curve, 64, 2024, edge=False)
    assert L.emu_pairs_check(P(pts), pts.shape[1], 64) == 0


def test_msm_bucket_sum_top_on_host(oracle):
    """The subset-sum top of the bucket sums (bucket_top_gather / _finish / _sum_gather): hand-over at 32 ... 4096 partial
    sums per window, chunks of 4 and 8, against the oracle and against the chunked levels alone (top = 1); with a
    work-group per PIECE of a sum (bucket_top_piece; the cooperative kernel's form: the plain sum of a 4096-item top in two
    pieces, and here also forced cuts of 2 / 4 and 1 / 2 pieces at every size) and per sum."""
    O = oracle
    L = _emu("BLS12_381")
    L.emu_msm_top_pieces.argtypes = [ctypes.c_uint]; L.emu_msm_top_pieces.restype = None
    n = 700
    pts, sc = recipe.msm_inputs(0, n, 4242, edge=False)
    exp = O.msm_affine(0, pts, sc, algo=0, param=4)
    try:
        for pieces in (1, 0, 24, 12):                               # automatic, per sum, forced "sb sp"
            L.emu_msm_top_pieces(pieces)
            for wb, K, top in ((9, 4, 0), (11, 8, 0), (12, 4, 0), (13, 4, 256), (13, 8, 32), (14, 8, 0), (15, 8, 0), (16, 8, 0), (16, 8, 1)):
                out = np.zeros(144, dtype=np.uint8)
                L.emu_msm(P(out), P(pts), 96, n, P(sc), 0, wb, 8, 4, K, 2, 1, None, top)
                assert (O.jac_to_affine(0, out) == exp).all(), (pieces, wb, K, top)
    finally:
        L.emu_msm_top_pieces(1)


@pytest.mark.parametrize("curve,feature", [(0, "BLS12_381"), (1, "BN254")])
def test_fixed_base_tables_on_host(oracle, curve, feature):
    """the fixed-base table work item (2^(off_j) * P_i, affine, one inversion per entry), the digit <-> table
    entry contract (entry w * n + i) and the one-bucket-set sum against the oracle's MSM"""
    O = oracle
    L = _emu(feature)
    fb = O.FP_BYTES[curve]
    for n, wb, flagged in ((1, 8, False), (37, 9, True), (200, 13, False), (64, 17, True)):
        pts, sc = recipe.msm_inputs(curve, n, 77 + n + wb, flagged=flagged)     # (edge cases: infinity, zero / r - 1 scalars)
        out = np.zeros(3 * fb, dtype=np.uint8)
        assert L.emu_fixed_base(P(out), P(pts), pts.shape[1], n, P(sc), wb) == 0
        assert (O.jac_to_affine(curve, out) == O.msm_affine(curve, pts, sc, algo=0, param=4)).all(), (n, wb)


@pytest.mark.parametrize("curve,feature,nr", [(2, "BLS12_381", 1), (3, "BN254", 1), (5, "BLS12_377", 5)])
def test_fp2_over_the_lazy_field_at_the_edge_of_its_bounds(oracle, curve, feature, nr):
    """ff/fp2x_dev.hpp operation by operation on the host, on representatives AT THE EDGE of the stated contract: the
    left operand of mul<KA> / the operand of sqr<KA> as x + (KA - 2) p (the largest normalised value below (KA - 1) p
    that is congruent to x), the right operand up to 15 p; sub / neg with the subtrahend at its bound.  Checked against
    Python integers: the result is congruent to the Montgomery product / difference, below 2 p for products (below
    (bound_a + KA) p for differences), with normalised limbs."""
    O = oracle
    L = _emu(feature, g2=True)
    p = O.FP_MODULUS[curve]
    LB = 28
    NL = (p.bit_length() + 8 + LB - 1) // LB
    R = 1 << (LB * NL)

    def limbs(v):                                   # normalised: limbs < 2^28, the top one takes what is left
        out = [(v >> (LB * j)) & ((1 << LB) - 1) for j in range(NL - 1)] + [v >> (LB * (NL - 1))]
        assert out[-1] < (1 << 31)
        return out

    def pack(els):                                  # list of (c0, c1) integers -> uint32 array of internal limbs
        return np.array([w for c0, c1 in els for w in limbs(c0) + limbs(c1)], dtype=np.uint32)

    def unpack(arr, n):
        a = arr.reshape(n, 2, NL).astype(object)
        val = lambda l: sum(int(l[j]) << (LB * j) for j in range(NL))
        for e in a:
            for c in e:
                assert all(int(c[j]) < (1 << LB) for j in range(NL - 1)), "limbs not normalised"
        return [(val(e[0]), val(e[1])) for e in a]

    rng = np.random.default_rng(7 + curve)
    rnd = lambda: int.from_bytes(rng.bytes(64), "little") % p
    n = 24
    for ka in (3, 6, 10, 13):
        edge = (ka - 2) * p
        xs = [(rnd() + edge, rnd() + edge) for _ in range(n)]
        xs[0] = (edge, edge); xs[1] = (p - 1 + edge, p - 1 + edge); xs[2] = (edge, p - 1 + edge)       # 0, -1 at the bound
        ys = [(rnd() + (k % 14) * p, rnd() + (13 - k % 14) * p) for k in range(n)]          # right operands up to 15 p
        ys[0] = (15 * p - 1, 15 * p - 1); ys[1] = (0, 0)
        A, B = pack(xs), pack(ys)
        out = np.zeros_like(A)
        assert L.emu_fp2x_op(0, ka, P(out), P(A), P(B), n) == 0
        for (a0, a1), (b0, b1), (c0, c1) in zip(xs, ys, unpack(out, n)):
            assert (c0 * R - (a0 * b0 - nr * a1 * b1)) % p == 0 and (c1 * R - (a0 * b1 + a1 * b0)) % p == 0, ("mul", ka)
            assert c0 < 2 * p and c1 < 2 * p, ("mul bound", ka, c0 // p, c1 // p)
        assert L.emu_fp2x_op(1, ka, P(out), P(A), P(A), n) == 0
        for (a0, a1), (c0, c1) in zip(xs, unpack(out, n)):
            assert (c0 * R - (a0 * a0 - nr * a1 * a1)) % p == 0 and (c1 * R - 2 * a0 * a1) % p == 0, ("sqr", ka)
            assert c0 < 2 * p and c1 < 2 * p, ("sqr bound", ka)
        # a - b with b at its bound (< (KA - 1) p), a anything normalised below 15 p
        assert L.emu_fp2x_op(2, ka, P(out), P(B), P(A), n) == 0
        for (b0, b1), (a0, a1), (c0, c1) in zip(ys, xs, unpack(out, n)):
            assert (c0 - (b0 - a0)) % p == 0 and (c1 - (b1 - a1)) % p == 0 and c0 == b0 + ka * p - a0 and c1 == b1 + ka * p - a1, ("sub", ka)
        assert L.emu_fp2x_op(3, ka, P(out), P(A), P(A), n) == 0
        for (a0, a1), (c0, c1) in zip(xs, unpack(out, n)):
            assert c0 == ka * p - a0 and c1 == ka * p - a1, ("neg", ka)


@pytest.mark.parametrize("curve,feature", [(2, "BLS12_381"), (3, "BN254"), (5, "BLS12_377")])
def test_g2_bucket_invariant_on_host(oracle, curve, feature):
    """ec/xyzzx2_dev.hpp: after every mixed addition, full addition and doubling of a chain the bucket satisfies what
    the next operation's template bounds assume -- all coordinates normalised, X < 9 p, Y < 5 p, ZZ, ZZZ < 2 p (both
    Fp2 components) -- and the point it represents is the oracle's (x = X / ZZ, y = Y / ZZZ checked through the MSM
    tests; here: the invariant)."""
    O = oracle
    L = _emu(feature, g2=True)
    p = O.FP_MODULUS[curve]
    LB = 28
    NL = (p.bit_length() + 8 + LB - 1) // LB
    n = 40
    pts, _ = recipe.msm_inputs(curve, n, 4242, ndistinct=n, flagged=True, edge=False)
    pts[1] = pts[0]                                             # step 1 is P - P: the infinity branch, step 2 restarts from infinity
    out = np.zeros((n + 3) * 4 * 2 * NL, dtype=np.uint32)
    k = L.emu_g2_chain(P(out), P(pts), pts.shape[1], n)
    assert k == n + 3
    img = out.reshape(k, 4, 2, NL)
    bounds = (9, 5, 2, 2)                                       # X, Y, ZZZ, ZZ
    for step in range(k):
        for coord in range(4):
            for comp in range(2):
                l = [int(v) for v in img[step, coord, comp]]
                assert all(v < (1 << LB) for v in l[:-1]), ("limbs not normalised", step, coord, comp)
                val = sum(v << (LB * j) for j, v in enumerate(l))
                assert val < bounds[coord] * p, ("bound", step, coord, comp, val // p)


def _emu_coop(feature):
    so = os.path.join(EMU, "libemu_coop_%s.so" % feature)
    src = os.path.join(EMU, "emu_coop.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "sppark_amd", "csrc")
    newest = max(os.stat(os.path.join(r, f)).st_mtime for r, _, fs in os.walk(csrc) for f in fs)
    newest = max(newest, os.stat(src).st_mtime)
    if not os.path.exists(so) or os.stat(so).st_mtime < newest:
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-DFEATURE_" + feature, "-o", so, src])
    L = ctypes.CDLL(so)
    vp, sz, ci, cu = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint
    L.emu_coop_ops.argtypes = [ci, vp, vp, vp, sz]
    L.emu_coop_tree.argtypes = [vp, vp, cu]
    L.emu_coop_reduce.argtypes = [vp, vp, cu, cu, ci, cu, vp, vp, vp, vp, vp, vp]
    L.emu_coop_from_std.argtypes = [vp, vp, sz]
    L.emu_coop_to_std.argtypes = [vp, vp, sz]
    L.emu_coop_record_words.restype = cu
    L.emu_coop_points_differ.argtypes = [vp, vp, sz]
    return L


@pytest.mark.parametrize("curve,feature", [(0, "BLS12_381"), (1, "BN254")])
def test_cooperative_point_ops_on_host(oracle, curve, feature):
    """ec/xyzz_coop.hpp and the cooperative record walk / tree of msm/msm_coop_kernels.hpp with a work-group emulated by
    256 host threads (tests/emu/emu_coop.cpp): two cooperative additions / doublings in a row against the serial
    formulas AND against the 32-bit-limb wire class, operands at infinity, equal operands (the doubling inside an
    addition), a count that is not a multiple of 64; the pairwise tree; one level of the record tree against the
    one-thread-per-item walk on records with runs, gaps and a run that crosses work items."""
    O = oracle
    L = _emu_coop(feature); W = _emu(feature)
    fb, n = O.FP_BYTES[curve], 96
    ofield = O.FIELD_BLS_FP if curve == 0 else O.FIELD_BN_FP
    one = O.field_op(ofield, 4, O.int_to_limbs(1, fb)).view(np.uint8)
    A = O.g1_gen_points(curve, n, 21)
    xa = np.zeros((n, 4 * fb), dtype=np.uint8); xa[:, :2 * fb] = A; xa[:, 2 * fb:3 * fb] = one; xa[:, 3 * fb:] = one
    tmp = np.zeros_like(xa); xb = np.zeros_like(xa)
    W.emu_xyzz_op(1, P(tmp), P(xa), P(O.g1_gen_points(curve, n, 22)), n); xa = tmp.copy()     # ZZ, ZZZ != 1
    W.emu_xyzz_op(1, P(xb), P(xa), P(O.g1_gen_points(curve, n, 23)), n)
    xb[0] = xa[0]; xa[3] = 0; xb[5] = 0; xa[7] = 0; xb[7] = 0
    words = int(L.emu_coop_record_words())
    ia = np.zeros((n, words), dtype=np.uint32); ib = np.zeros_like(ia)
    L.emu_coop_from_std(P(ia), P(xa), n); L.emu_coop_from_std(P(ib), P(xb), n)

    def std(rec):
        out = np.zeros((rec.shape[0], 4 * fb), dtype=np.uint8)
        L.emu_coop_to_std(P(out), P(np.ascontiguousarray(rec)), rec.shape[0])
        return out
    r1 = np.zeros_like(xa); r2 = np.zeros_like(xa)
    for coop_op, serial_op, wire_op, operand in ((0, 2, 0, xb), (1, 3, 3, None)):
        got = np.zeros_like(ia); ser = np.zeros_like(ia)
        L.emu_coop_ops(coop_op, P(got), P(ia), P(ib), n); L.emu_coop_ops(serial_op, P(ser), P(ia), P(ib), n)
        W.emu_xyzz_op(wire_op, P(r1), P(xa), P(operand) if operand is not None else None, n)
        W.emu_xyzz_op(wire_op, P(r2), P(r1), P(operand) if operand is not None else None, n)
        assert (std(got) == std(ser)).all() and (std(got) == r2).all(), coop_op
    # the tree: a power-of-two count of points, some at infinity
    pts = np.concatenate([ia, ib])[:256] if 2 * n >= 256 else None
    if pts is None:
        pts = np.concatenate([ia, ib, ia])[:256]
    pts = np.ascontiguousarray(pts)
    for count in (2, 64, 256):
        out = np.zeros((2, words), dtype=np.uint32)
        L.emu_coop_tree(P(out), P(pts), count)
        assert L.emu_coop_points_differ(P(out[:1].copy()), P(out[1:].copy()), 1) == 0, count    # (projective: another order of summation)
    # one level of the record tree: runs of 1..9 records with one key, NONE gaps, fan-in 4 and 8, last level too
    NONE = 0xffffffff
    rng = np.random.default_rng(5)
    keys = []
    k = 0
    while len(keys) < 700:
        run = int(rng.integers(1, 10))
        keys += [k] * run
        if rng.random() < 0.4:
            keys += [NONE] * int(rng.integers(1, 3))
        k += 1
    keys = np.array(keys[:700], dtype=np.uint32)
    nb = int(keys[keys != NONE].max()) + 1
    recs = np.ascontiguousarray(np.concatenate([ia, ib] * 4)[:700])
    for fan, nrec, last in ((4, 700, 0), (8, 700, 0), (4, 4, 1), (8, 7, 1)):
        nthreads = (nrec + fan - 1) // fan
        res = []
        for _ in range(2):
            res.append([np.zeros((nb, words), dtype=np.uint32), np.full(2 * nthreads, 0xabcdef01, dtype=np.uint32), np.zeros((2 * nthreads, words), dtype=np.uint32)])
        L.emu_coop_reduce(P(keys), P(recs), nrec, fan, last, nb, P(res[0][0]), P(res[0][1]), P(res[0][2]), P(res[1][0]), P(res[1][1]), P(res[1][2]))
        assert (res[0][1] == res[1][1]).all(), (fan, nrec, "keys")
        # (same order of additions in both walks: the records agree coordinate by coordinate)
        assert (std(res[0][0]) == std(res[1][0])).all(), (fan, nrec, "buckets")
        live = res[0][1] != NONE if not last else np.zeros(2 * nthreads, dtype=bool)
        assert (std(res[0][2][live]) == std(res[1][2][live])).all(), (fan, nrec, "records")


def _emu_bounds(feature, g2=False):
    so = os.path.join(EMU, "libemu_bounds_%s%s.so" % (feature, "_G2" if g2 else ""))
    src = os.path.join(EMU, "emu_bounds.cpp")
    csrc = os.path.join(os.path.dirname(HERE), "sppark_amd", "csrc")
    newest = max(os.stat(os.path.join(r, f)).st_mtime for r, _, fs in os.walk(csrc) for f in fs)
    newest = max(newest, os.stat(src).st_mtime)
    if not os.path.exists(so) or os.stat(so).st_mtime < newest:
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-O1", "-std=c++17", "-fPIC", "-shared",
                               "-DFEATURE_" + feature] + (["-DSPPARK_G2"] if g2 else []) + ["-o", so, src])
    L = ctypes.CDLL(so)
    L.emu_bounds_run.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    return L


@pytest.mark.parametrize("curve,feature,g2", [(0, "BLS12_381", False), (1, "BN254", False), (4, "BLS12_377", False), (6, "PALLAS", False),
                                              (2, "BLS12_381", True), (3, "BN254", True), (5, "BLS12_377", True)])
def test_lazy_field_contracts_are_machine_checked(oracle, curve, feature, g2):
    """ff/montx_dev.hpp with -DSPPARK_TRACK_BOUNDS (tests/emu/emu_bounds.cpp): every field value carries its CLAIMED bounds
    (value < K p, limbs <= B 2^LB), every operation checks its operands against its stated precondition and derives its
    result's claim by the rule at its definition.  Every point operation of the G1 / G2 bucket classes -- mixed addition
    with and without negation, full addition, doubling, the interleaved-pair forms, the doubling and equal-operand
    branches, conversion in and out, chains -- is started from the LOOSEST operands the memory invariants allow: no
    precondition is violated on the way and every result is inside the invariants again.  For all inputs, not a sample.
    And the checker itself reports deliberately broken contracts."""
    L = _emu_bounds(feature, g2)
    pts, _sc = recipe.msm_inputs(curve, 12, 2024, edge=False, flagged=False)
    msg = ctypes.create_string_buffer(512)
    nv = L.emu_bounds_run(P(pts), pts.shape[1], pts.shape[0], msg, 512)
    assert nv == 0, (nv, msg.value.decode())
    assert L.emu_bounds_selftest() == 5


def _emu_tracked(src_name, tag, feature, g2=False):
    """tests/emu/<src_name> built with -DSPPARK_TRACK_BOUNDS (the field values carry their claimed bounds)"""
    so = os.path.join(EMU, "libemu_%s_bounds_%s%s.so" % (tag, feature, "_G2" if g2 else ""))
    src = os.path.join(EMU, src_name)
    csrc = os.path.join(os.path.dirname(HERE), "sppark_amd", "csrc")
    newest = max(os.stat(os.path.join(r, f)).st_mtime for r, _, fs in os.walk(csrc) for f in fs)
    newest = max(newest, os.stat(src).st_mtime)
    if not os.path.exists(so) or os.stat(so).st_mtime < newest:
        if not os.path.exists(HIPCC):
            pytest.skip("hipcc not available")
        subprocess.check_call([HIPCC, "-x", "hip", "--cuda-host-only", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DSPPARK_TRACK_BOUNDS",
                               "-DFEATURE_" + feature] + (["-DSPPARK_G2"] if g2 else []) + ["-o", so, src])
    return ctypes.CDLL(so)


@pytest.mark.parametrize("curve,feature", [(0, "BLS12_381"), (1, "BN254")])
def test_cooperative_operations_keep_the_field_contracts(oracle, curve, feature):
    """coop_add / coop_dbl (ec/xyzz_coop.hpp: the four-wave point operations of the MSM tail, a DEFAULT path) with the
    bound tracking of test_lazy_field_contracts_are_machine_checked: from the loosest operands the bucket invariants
    allow, through the exchange area (the claims travel with the limbs), equal operands included -- no precondition
    violated, every result inside the invariants."""
    L = _emu_tracked("emu_coop.cpp", "coop", feature)
    L.emu_coop_bounds.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    pts, _sc = recipe.msm_inputs(curve, 70, 77, edge=False, flagged=False)
    msg = ctypes.create_string_buffer(512)
    nv = L.emu_coop_bounds(P(pts), pts.shape[1], pts.shape[0], msg, 512)
    assert nv == 0, (nv, msg.value.decode())


@pytest.mark.parametrize("curve,feature", [(2, "BLS12_381"), (3, "BN254"), (5, "BLS12_377")])
def test_g2_wave_pair_bucket_keeps_the_field_contracts(oracle, curve, feature):
    """g2c_bucket::madd (ec/xyzz2_coop.hpp, the default-off G2 accumulation by wave pairs) under the same bound tracking:
    restart, plain step, the same point again (the cooperative doubling), chains -- from the loosest operands."""
    L = _emu_tracked("emu_msm.cpp", "g2c", feature, g2=True)
    L.emu_g2c_bounds.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    pts, _sc = recipe.msm_inputs(curve, 80, 78, edge=False, flagged=False)
    msg = ctypes.create_string_buffer(512)
    nv = L.emu_g2c_bounds(P(pts), pts.shape[1], pts.shape[0], msg, 512)
    assert nv == 0, (nv, msg.value.decode())
