"""CPU: pin the oracle (oracle/) against Python big-ints, the KATs of SURVEY
Appendix A, the golden vectors and -- when built -- the reference itself."""
import json
import os

import numpy as np
import pytest

import recipe

HERE = os.path.dirname(os.path.abspath(__file__))


def _rand_field(rng, p, nbytes, n):
    vals = [int.from_bytes(rng.bytes(nbytes + 8), "little") % p for _ in range(n)]
    vals[:4] = [0, 1, p - 1, p - 2]
    return vals


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_field_ops_vs_python(oracle, field):
    O = oracle
    p = {0: O.FP_MODULUS[0], 1: O.FR_MODULUS[0], 2: O.FP_MODULUS[1], 3: O.FR_MODULUS[1]}[field]
    nb = 48 if field == 0 else 32
    R = 1 << (8 * nb)
    rinv = pow(R, p - 2, p)
    rng = np.random.default_rng(field)
    a, b = _rand_field(rng, p, nb, 64), _rand_field(rng, p, nb, 64)[::-1]
    for x, y in zip(a, b):
        lx, ly = O.int_to_limbs(x, nb), O.int_to_limbs(y, nb)
        assert O.limbs_to_int(O.field_op(field, 0, lx, ly)) == (x + y) % p
        assert O.limbs_to_int(O.field_op(field, 1, lx, ly)) == (x - y) % p
        assert O.limbs_to_int(O.field_op(field, 2, lx, ly)) == x * y * rinv % p
        assert O.limbs_to_int(O.field_op(field, 7, lx)) == x * x * rinv % p
        assert O.limbs_to_int(O.field_op(field, 4, lx)) == x * R % p
        assert O.limbs_to_int(O.field_op(field, 5, lx)) == x * rinv % p
        assert O.limbs_to_int(O.field_op(field, 6, lx)) == (-x) % p
        inv = O.limbs_to_int(O.field_op(field, 3, lx))                  # Montgomery inverse
        assert (x * inv * rinv * rinv) % p == (1 if x else 0)


def test_reference_constants(oracle):
    """R, R^2, -1/p of the parameter tables (ff/bls12-381.hpp, ff/alt_bn128.hpp)."""
    O = oracle
    for field, p, nb in ((0, O.FP_MODULUS[0], 48), (1, O.FR_MODULUS[0], 32), (2, O.FP_MODULUS[1], 32), (3, O.FR_MODULUS[1], 32)):
        one = O.limbs_to_int(O.field_op(field, 4, O.int_to_limbs(1, nb)))
        assert one == (1 << (8 * nb)) % p
    assert 0x1ffffedc * pow(1 << 32, O.BB31_P - 2, O.BB31_P) % O.BB31_P == 137
    assert pow(7, (O.GL64_P - 1) >> 32, O.GL64_P) == 0x185629dcda58878c
    assert O.lib().oracle_gl64_root(24) == 0x86cdcc31c307e171
    assert O.lib().oracle_gl64_root(1) == O.GL64_P - 1


def test_msm_kat_30G(oracle):
    """SURVEY Appendix A.4."""
    O = oracle
    G = O.g1_generator(O.BLS12_381)
    pts = np.stack([O.g1_mul(O.BLS12_381, G, i) for i in range(1, 5)])
    sc = np.stack([np.frombuffer(int(i).to_bytes(32, "little"), dtype=np.uint8) for i in range(1, 5)])
    x_limbs = [0x81d35a7c0d45fea7, 0x1d72ddc201bd6796, 0x26fca92118a3cb77, 0x412b51bb870f2a3a, 0x67640ebd91ea6ce3, 0x13d09f545c6c9d60]
    for algo, param in ((0, 0), (0, 8), (1, 0), (2, 10), (2, 16)):
        a = O.msm_affine(O.BLS12_381, pts, sc, algo=algo, param=param)
        assert list(a[:48].view(np.uint64)) == x_limbs
    xc = O.limbs_to_int(O.field_op(O.FIELD_BLS_FP, 5, a[:48].view(np.uint64)))
    assert xc == int("0d84464b3966ec5bede84aa487facfca7823af383715078da03b387cc2f5d5597cdd7d025aa07db00a38b953bdeb6e3f", 16)
    assert O.g1_on_curve(O.BLS12_381, a)


def test_msm_golden(oracle):
    """Every MSM golden vector (expectations produced by the reference's own
    msm/pippenger.hpp, tests/golden/make_golden.py) through all three oracle
    evaluators."""
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "msm_golden.json"))):
        curve = O.CURVE_ID[c["curve"]]
        fb = O.FP_BYTES[curve]
        stride = 2 * fb + 8 if c["flagged"] else 2 * fb
        if "points" in c:
            pts = np.frombuffer(bytes.fromhex(c["points"]), dtype=np.uint8).reshape(c["n"], stride)
            sc = np.frombuffer(bytes.fromhex(c["scalars"]), dtype=np.uint8).reshape(c["n"], 32)
        else:
            pts, sc = recipe.msm_inputs(curve, c["n"], c["seed"], c["ndistinct"], c["flagged"])
        exp = np.frombuffer(bytes.fromhex(c["expect_affine"]), dtype=np.uint8)
        algos = [(0, 0), (0, 4), (2, 9)] + ([(1, 0)] if c["n"] <= 1024 else [])
        if c["n"] > 4096:
            algos = [(0, 8)]
        for algo, param in algos:
            got = O.msm_affine(curve, pts, sc, algo=algo, param=param)
            assert (got[:2 * fb] == exp).all(), (c["curve"], c["n"], algo)


@pytest.mark.parametrize("fname,g2", [("msm_golden.json", False), ("msm_g2_golden.json", True)])
def test_msm_golden_vs_independent_python_group_law(oracle, fname, g2):
    """Second, independent pin of the MSM expectations: tests/golden/pygroup.py (pure-Python
    big-int affine arithmetic, decodes the wire bytes itself, shares nothing with oracle/) re-derives
    every golden case whose bytes are stored (all edge cases, both curves, G1 and G2) and one
    n = 1000 case per file; it must equal the committed expectation (= the reference build's
    output) AND the restatement."""
    import pygroup
    O = oracle
    big_done = set()
    for c in json.load(open(os.path.join(HERE, "golden", fname))):
        curve = (O.CURVE_ID_G2 if g2 else O.CURVE_ID)[c["curve"]]
        fb = O.FP_BYTES[curve]
        stride = 2 * fb + 8 if c["flagged"] else 2 * fb
        if "points" in c:
            praw, sraw = bytes.fromhex(c["points"]), bytes.fromhex(c["scalars"])
        elif c["n"] == 1000 and c["curve"] not in big_done and not g2:
            big_done.add(c["curve"])
            pts, sc = recipe.msm_inputs(curve, c["n"], c["seed"], c["ndistinct"], c["flagged"])
            praw, sraw = pts.tobytes(), sc.tobytes()
        else:
            continue
        got = pygroup.msm_affine_bytes(c["curve"], g2, praw, stride, c["flagged"], sraw)
        assert got.hex() == c["expect_affine"], (fname, c["curve"], c["n"], c.get("name"))
        pts = np.frombuffer(praw, dtype=np.uint8).reshape(c["n"], stride)
        sc = np.frombuffer(sraw, dtype=np.uint8).reshape(c["n"], 32)
        assert O.msm_affine(curve, pts, sc, algo=0, param=0).tobytes() == got


def test_msm_vs_reference_build(oracle):
    """Restatement == the reference's own template, on fresh random inputs."""
    O = oracle
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    for curve in (O.BLS12_381, O.BN254):
        for n, thr in ((5, 0), (100, 0), (100, 3), (777, 8)):
            pts, sc = recipe.msm_inputs(curve, n, 99 + n)
            assert (O.ref_msm_affine(curve, pts, sc, thr) == O.msm_affine(curve, pts, sc, algo=0, param=thr)).all()
        pts, sc = recipe.msm_inputs(curve, 64, 5)
        mont = np.zeros_like(sc)
        fld = O.FIELD_BLS_FR if curve == O.BLS12_381 else O.FIELD_BN_FR
        for i in range(64):
            mont[i] = O.field_op(fld, 4, sc[i].view(np.uint64)).view(np.uint8)
        assert (O.ref_msm_affine(curve, pts, mont, 0, mont=True) == O.msm_affine(curve, pts, sc)).all()
        assert (O.msm_affine(curve, pts, mont, mont=True) == O.msm_affine(curve, pts, sc)).all()


def test_msm_g2_golden_and_reference_build(oracle):
    """G2 (coordinates in Fp2): the restatement against (i) the committed vectors produced by the
    reference's own templates instantiated over Fp2 (+ the 30*G2 KAT computed by an independent
    Python big-int group law), (ii) the reference build itself on fresh inputs, (iii) itself
    through the naive and the signed-window algorithms."""
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "msm_g2_golden.json"))):
        curve = O.CURVE_ID_G2[c["curve"]]
        fb = O.FP_BYTES[curve]
        stride = 2 * fb + 8 if c["flagged"] else 2 * fb
        if "points" in c:
            pts = np.frombuffer(bytes.fromhex(c["points"]), dtype=np.uint8).reshape(c["n"], stride).copy()
            sc = np.frombuffer(bytes.fromhex(c["scalars"]), dtype=np.uint8).reshape(c["n"], 32).copy()
        else:
            pts, sc = recipe.msm_inputs(curve, c["n"], c["seed"], c["ndistinct"], c["flagged"])
        exp = np.frombuffer(bytes.fromhex(c["expect_affine"]), dtype=np.uint8)
        assert (O.msm_affine(curve, pts, sc, algo=0, param=0) == exp).all(), (c["curve"], c["n"])
        if c["n"] <= 1000:
            assert (O.msm_affine(curve, pts, sc, algo=2, param=7) == exp).all()
        if c["n"] <= 33:
            assert (O.msm_affine(curve, pts, sc, algo=1) == exp).all()
    for curve in (O.BLS12_381_G2, O.BN254_G2):
        g = O.g1_generator(curve)
        assert O.g1_on_curve(curve, g)
        if O.ref_available():
            for n, thr in ((5, 0), (100, 3), (300, 8)):
                pts, sc = recipe.msm_inputs(curve, n, 199 + n, flagged=True)
                assert (O.ref_msm_affine(curve, pts, sc, thr) == O.msm_affine(curve, pts, sc, algo=0, param=thr)).all()


def test_msm_empty(oracle):
    O = oracle
    out = O.msm(O.BLS12_381, np.zeros((0, 96), dtype=np.uint8), np.zeros((0, 32), dtype=np.uint8))
    assert (out[96:] == 0).all()


def test_ntt_golden(oracle):
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "ntt_golden.json"))):
        dt = np.uint32 if c["field"] == "bb31" else np.uint64
        x = np.frombuffer(bytes.fromhex(c["input"]), dtype=dt)
        e = np.frombuffer(bytes.fromhex(c["expect"]), dtype=dt)
        if c["field"] in O.CURVE_ID:
            curve = O.CURVE_ID[c["field"]]
            got = O.ntt_fr(curve, x.reshape(-1, 4), c["order"], c["direction"], c["type"]).reshape(-1)
        else:
            got = (O.ntt_gl64 if c["field"] == "gl64" else O.ntt_bb31)(x, c["order"], c["direction"], c["type"])
        assert (got == e).all(), c


def test_lde_golden(oracle):
    """oracle LDE (restating NTT::LDE_aux, ntt/ntt.cuh:283-336) against the definition-level
    big-int vectors: coset evaluations of the interpolating polynomial + its coefficients;
    and the reference's reading of LDE as iNTT -> coset NTT on the zero-padded coefficients."""
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "lde_golden.json"))):
        dt = np.uint32 if c["field"] == "bb31" else np.uint64
        w = 4 if c["field"] in O.CURVE_ID else 1
        x = np.frombuffer(bytes.fromhex(c["input"]), dtype=dt).reshape(-1, w)
        got, aux = O.lde(c["field"], x, c["lg_blowup"], want_aux=True)
        assert (got.reshape(-1) == np.frombuffer(bytes.fromhex(c["expect"]), dtype=dt)).all(), c
        assert (aux.reshape(-1) == np.frombuffer(bytes.fromhex(c["aux"]), dtype=dt)).all(), c
        # same thing through the plain transforms: pad the coefficients, coset NTT NN
        if c["lg"] + c["lg_blowup"] > 0 and w == 1:
            pad = np.zeros(x.shape[0] << c["lg_blowup"], dtype=dt)
            pad[:x.shape[0]] = aux
            f = O.ntt_gl64 if c["field"] == "gl64" else O.ntt_bb31
            assert (f(pad, O.NN, O.FORWARD, O.COSET) == got).all()
        # expand / powers helpers
        e = O.lde_expand(c["field"], x, c["lg_blowup"]).reshape(-1, w)
        assert (e[::1 << c["lg_blowup"]] == x).all() and int((e != 0).sum()) == int((x != 0).sum())


def test_wide_ntt_roots_match_reference_tables(oracle):
    """forward_roots_of_unity[k] of ntt/parameters/{bls12_381,alt_bn128}.h, every k
    (the tables are read here, in the build container, only; nothing is copied)."""
    import re
    O = oracle
    if not os.path.isdir("/root/reference"):
        pytest.skip("no /root/reference here")
    for curve, name, S in ((O.BLS12_381, "bls12_381", 32), (O.BN254, "alt_bn128", 28)):
        txt = open("/root/reference/ntt/parameters/%s.h" % name).read()
        m = re.search(r"forward_roots_of_unity\[S \+ 1\] = \{(.*?)\};", txt, re.S)
        ents = re.findall(r"FR_T\(vec256, (0x[0-9a-f]+)u, (0x[0-9a-f]+)u, (0x[0-9a-f]+)u, (0x[0-9a-f]+)u", m.group(1))
        assert len(ents) == S + 1
        for k in range(S + 1):
            assert [int(v, 16) for v in ents[k]] == [int(v) for v in O.fr_root(curve, k)], (name, k)


def test_ntt_kat_survey_a3(oracle):
    O = oracle
    x = np.arange(1, 9, dtype=np.uint64)
    nn = [0x24, 0xfffc03ff03fffbfd, 0xfffbfffefffffffd, 0x0004040003fffbfc,
          0xfffffffefffffffd, 0xfffbfbfefc0003fd, 0x0003fffffffffffc, 0x0003fbfffc0003fc]
    assert list(O.ntt_gl64(x, O.NN)) == nn
    assert list(O.ntt_naive_gl64(x)) == nn
    assert list(O.ntt_gl64(x, O.NR)) == [nn[int(format(i, "03b")[::-1], 2)] for i in range(8)]
    assert int(O.ntt_gl64(x, O.NN, O.FORWARD, O.COSET)[0]) == 0x72d77c


@pytest.mark.parametrize("field", ["gl64", "bb31"])
def test_ntt_properties(oracle, field):
    """The reference's own test shapes (poc/ntt-cuda/tests/ntt.rs:9-78):
    NN == RR, iNTT(NTT(v)) == v in NN/RR, iNTT_RN(NTT_NR(v)) == v."""
    O = oracle
    f = O.ntt_gl64 if field == "gl64" else O.ntt_bb31
    naive = O.ntt_naive_gl64 if field == "gl64" else O.ntt_naive_bb31
    for lg in range(1, 11):
        v = recipe.ntt_input(field, lg, lg)
        nn = f(v, O.NN)
        assert (nn == f(v, O.RR)).all()
        if lg <= 8:
            assert (nn == naive(v)).all()
        assert (f(nn, O.NN, O.INVERSE) == v).all()
        assert (f(f(v, O.RR), O.RR, O.INVERSE) == v).all()
        assert (f(f(v, O.NR), O.RN, O.INVERSE) == v).all()
        for order in (O.NN, O.RR):
            assert (f(f(v, order, O.FORWARD, O.COSET), order, O.INVERSE, O.COSET) == v).all()
        assert (f(f(v, O.NR, O.FORWARD, O.COSET), O.RN, O.INVERSE, O.COSET) == v).all()


def test_poly_golden(oracle):
    """oracle/poly.hpp against the definition-level Python big-int vectors (the reference has no CPU
    version of polynomial/*.cuh and no tests for them)."""
    O = oracle
    for c in json.load(open(os.path.join(HERE, "golden", "poly_golden.json"))):
        f = c["field"]
        dt = np.uint32 if f in ("bb31", "m31", "bb31x4") else np.uint64
        w = 4 if (f in O.CURVE_ID or f == "bb31x4") else 1
        arr = lambda key: np.frombuffer(bytes.fromhex(c[key]), dtype=dt).reshape(-1, w).squeeze(-1) if w == 1 else \
            np.frombuffer(bytes.fromhex(c[key]), dtype=dt).reshape(-1, w)
        coeffs, z = arr("coeffs"), arr("z")
        assert (O.div_by_x_minus_z(f, coeffs, z) == arr("div")).all(), (f, c["len"])
        assert (O.div_by_x_minus_z(f, coeffs, z, rotate=True) == arr("div_rotate")).all(), (f, c["len"])
        if "prefix_add" in c:
            assert (O.prefix_op(f, coeffs, 0) == arr("prefix_add")).all(), (f, c["len"])
            assert (O.prefix_op(f, coeffs, 1) == arr("prefix_mul")).all(), (f, c["len"])
            assert (O.poly_evaluate(f, coeffs, arr("xs")) == arr("evaluate")).all(), (f, c["len"])


def test_root_convention_variants(oracle):
    """-DGOLDILOCKS_PLONKY2 / -DBABY_BEAR_CANONICAL: the oracle's switchable conventions against the
    first entries of the reference's own tables (ntt/parameters/goldilocks.h:12-19,
    ntt/parameters/baby_bear.h:14-18: forward_roots_of_unity[0..7] / [0..4]) and its group generators
    (goldilocks.h:9-10, baby_bear.h:9-10)."""
    O = oracle
    L = O.lib()
    try:
        O.set_root_conventions(True, True)
        assert [L.oracle_gl64_root(k) for k in range(8)] == [1, 0xffffffff00000000, 0x0001000000000000, 0x0000000001000000,
                                                              0x1000, 0x40, 0x8, 0x000001fffdfffe00]
        assert L.oracle_gl64_root(32) == 0x64fdd1a46201e246
        assert [L.oracle_bb31_root(k) for k in range(5)] == [0x0ffffffe, 0x68000003, 0x1c38d511, 0x3d85298f, 0x5f06e481]
        assert L.oracle_bb31_root(27) == 0x57fab6ee
        p = O.BB31_P
        assert 31 * 0x03def7be % p == 1 and 0xc65c18b67785d900 * 0xb1ddc963fcd29ccc % O.GL64_P == 1     # group_gen * group_gen_inverse
        # coset transform uses the variant's generator: x = delta_1 -> X[k] = g * w^k
        x = np.zeros(8, dtype=np.uint64); x[1] = 1
        y = O.ntt_gl64(x, O.NN, O.FORWARD, O.COSET)
        assert int(y[0]) == 0xc65c18b67785d900 and int(y[1]) == 0xc65c18b67785d900 * 0x0000000001000000 % O.GL64_P
    finally:
        O.set_root_conventions(False, False)
    assert L.oracle_gl64_root(32) == 0x185629dcda58878c and L.oracle_bb31_root(27) == 0x1ffffedc


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only in the build container")
def test_reference_ntt_build_recipe(oracle):
    """oracle/Makefile `ref_ntt`: the reference's own NTT builds for gfx950 through its HIP path from the sources where
    they lie -- nine libraries, each exporting the reference's compute_ntt (poc/ntt-cuda/cuda/ntt_api.cu:25-36) and the
    forwarders of oracle/ref_ntt_shim.cu; they hold gfx950 device code and none of sppark_amd's symbols.  (They RUN on
    the GPU box only: tests/test_ntt_vs_reference_gpu.py.)"""
    import subprocess
    O = oracle
    here = os.path.dirname(os.path.abspath(O.__file__))
    subprocess.check_call(["make", "-s", "-C", here, "ref_ntt"])
    for field in O.REF_NTT_FIELDS:
        assert O.ref_ntt_available(field), field
        so = os.path.join(here, "_ref", "libref_ntt_%s.so" % field)
        syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
        for s in ("compute_ntt", "cuda_available", "ref_ntt_dev", "ref_ntt_dev_timed", "ref_lde_aux", "ref_ntt_elem_bytes"):
            assert (" T " + s + "\n") in syms, (field, s)
        assert "sppark_amd" not in syms and "sppark_ntt" not in syms     # (the reference has a class sppark_error of its own)
        fat = subprocess.run(["strings", "-n", "6", so], capture_output=True, text=True).stdout
        assert "gfx950" in fat, field
    # the recipe is in .gitignore (outputs never committed) and NOT in .gpurunignore (they travel to the GPU box)
    root = os.path.dirname(here)
    assert "oracle/_ref/" in open(os.path.join(root, ".gitignore")).read()
    ignore = os.path.join(root, ".gpurunignore")
    assert not os.path.exists(ignore) or "oracle/_ref" not in open(ignore).read()


def test_oracle_field_equals_the_reference_device_field(oracle):
    """The ORACLE's field (oracle/ff.hpp: SOS Montgomery on 64-bit limbs, CPU) against outputs of the REFERENCE's own device
    field classes -- fp_t / fr_t over ff/mont_t.hip (ff/bls12-381.hpp:63-83 and the other curves' headers), built for gfx950
    and run on an MI355X by tests/golden/make_ref_field_golden.py, which recorded operands and results in
    tests/golden/ref_field_golden.json: + - * sqr to() from() on edge-heavy operands, base and scalar field of five curves.
    (The DEVICE classes are held against the same reference code live on the GPU: tests/test_field_vs_reference_gpu.py.)"""
    O = oracle
    with open(os.path.join(HERE, "golden", "ref_field_golden.json")) as f:
        gold = json.load(f)
    curve_id = {"bls12_381": O.BLS12_381, "bn254": O.BN254, "bls12_377": O.BLS12_377, "pallas": O.PALLAS, "vesta": O.VESTA}
    oracle_op = {"add": 0, "sub": 1, "mul": 2, "sqr": 7, "to": 4, "from": 5}      # oracle_capi.cpp's op numbers
    assert gold["ops"] == ["add", "sub", "mul", "sqr", "to", "from"] and len(gold["cases"]) == 10
    for c in gold["cases"]:
        curve = curve_id[c["curve"]]
        fid = (O.FP_FIELD_ID if c["field"] == "fp" else O.FR_FIELD_ID)[curve]
        nb, n = c["bytes"], c["n"]
        p = int(c["modulus"], 16)
        assert p == (O.FP_MODULUS if c["field"] == "fp" else O.FR_MODULUS)[curve]
        a = np.frombuffer(bytes.fromhex(c["a"]), dtype=np.uint8).reshape(n, nb)
        b = np.frombuffer(bytes.fromhex(c["b"]), dtype=np.uint8).reshape(n, nb)
        R = 1 << (8 * nb); Rinv = pow(R, p - 2, p)
        for name, oop in oracle_op.items():
            exp = np.frombuffer(bytes.fromhex(c["expect"][name]), dtype=np.uint8).reshape(n, nb)
            for i in range(n):
                got = O.field_op(fid, oop, a[i].copy().view(np.uint64), b[i].copy().view(np.uint64)).view(np.uint8)
                assert (got == exp[i]).all(), (c["curve"], c["field"], name, i)
            # ... and the recorded outputs are what the operation means (big integers): the fixture is not self-referential
            for i in (0, 3, 5, 14, n - 1):
                x, y, e = (int.from_bytes(v[i].tobytes(), "little") for v in (a, b, exp))
                want = {"add": (x + y) % p, "sub": (x - y) % p, "mul": x * y * Rinv % p, "sqr": x * x * Rinv % p,
                        "to": x * R % p, "from": x * Rinv % p}[name]
                assert e == want, (c["curve"], c["field"], name, i)


def test_oracle_ntt_equals_the_reference_build_vectors(oracle):
    """The ORACLE's NTT restatement (oracle/ntt.hpp) against outputs of the REFERENCE's own compute_ntt -- its HIP build for gfx950,
    run on an MI355X by tests/golden/make_ref_ntt_golden.py and recorded in tests/golden/ref_ntt_golden.json: every order x
    direction x type at 2^1 .. 2^5, nine field libraries incl. both compile-time root conventions.  (The `-m gpu` suite applies
    the same pin live and at every size: tests/test_ntt_vs_reference_gpu.py.)"""
    O = oracle
    with open(os.path.join(HERE, "golden", "ref_ntt_golden.json")) as f:
        gold = json.load(f)
    assert len(gold["cases"]) == 31
    try:
        for c in gold["cases"]:
            O.set_root_conventions(goldilocks_plonky2=c["lib"] == "gl64_plonky2", baby_bear_canonical=c["lib"] == "bb31_canonical")
            x = np.frombuffer(bytes.fromhex(c["input"]), dtype=np.dtype(c["dtype"]))
            if c["kind"] == "gl64":
                f = O.ntt_gl64
            elif c["kind"] == "bb31":
                f = O.ntt_bb31
            else:
                x = x.reshape(-1, 4)
                f = (lambda curve: (lambda a, o, d, t: O.ntt_fr(curve, a, o, d, t)))(O.CURVE_ID[c["kind"]])
            assert x.shape[0] == 1 << c["lg"]
            assert len(c["expect"]) == 16
            for key, hexed in c["expect"].items():
                order, direction, typ = (int(ch) for ch in key)
                exp = np.frombuffer(bytes.fromhex(hexed), dtype=x.dtype).reshape(x.shape)
                assert (f(x, order, direction, typ) == exp).all(), (c["lib"], c["lg"], key)
    finally:
        O.set_root_conventions(False, False)
