"""GPU parity tests for the NTT hot path, through the C ABI.  Outputs are unique
bit patterns (canonical u64 / Montgomery u32 < p), so equality is exact."""
import json
import os

import numpy as np
import pytest

import recipe

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ["gl64", "bb31"]
WIDE = ["bls12_381", "bn254", "bls12_377", "pallas", "vesta"]


def _oracle_fn(O, field):
    if field in WIDE:
        curve = O.CURVE_ID[field]
        return lambda x, order, direction, typ: O.ntt_fr(curve, x, order, direction, typ)
    return O.ntt_gl64 if field == "gl64" else O.ntt_bb31


def test_device_small_field_ops(libs):
    """element-wise gl64 / bb31 arithmetic on the GPU vs Python big-ints, edge-heavy
    inputs, including chained and divergent use (hand-written carry chains)."""
    import random
    from sppark_amd import ffi
    random.seed(3)
    for field, P, dt in (("gl64", 0xffffffff00000001, np.uint64), ("bb31", 0x78000001, np.uint32)):
        L = ffi.load_devtest(field)
        M32 = 0xffffffff
        R = 1 if field == "gl64" else pow(1 << 32, P - 2, P)         # bb31 values are Montgomery residues

        def rnd():
            r = random.random()
            if field == "bb31":
                return random.choice([0, 1, P - 1, P - 2]) if r < 0.2 else random.getrandbits(32) % P
            if r < 0.2:
                return random.choice([0, 1, P - 1, P - 2, M32, M32 << 32, (M32 << 32) - 1, 1 << 32, (1 << 32) - 1, (1 << 32) + 1])
            if r < 0.4:
                return ((M32 << 32) | random.getrandbits(32)) % P
            if r < 0.6:
                return random.getrandbits(32)
            return random.getrandbits(64) % P
        n = 1 << 13
        a = np.array([rnd() for _ in range(n)], dtype=dt); b = np.array([rnd() for _ in range(n)], dtype=dt)
        for op in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            if field == "bb31" and op == 3:
                continue
            out = np.zeros(n, dtype=dt)
            ffi.check(L, L.sppark_devtest_small_field_op(op, out.ctypes.data, a.ctypes.data, b.ctypes.data, n))
            for i in range(n):
                x, y = int(a[i]), int(b[i])
                mul = lambda u, v: u * v * R % P
                if op in (0, 7): e = (x + y) % P
                elif op in (1, 8): e = (x - y) % P
                elif op == 2: e = mul(x, y)
                elif op == 3: e = x * pow(2, (y & 0xff) % 192, P) % P
                elif op == 4:
                    e = x
                    for _ in range(12): e = mul(e, e)
                elif op == 5:
                    e, bb, k = (1 if field == "gl64" else (1 << 32) % P), x, y & 0xffff
                    while k:
                        if k & 1: e = mul(e, bb)
                        bb = mul(bb, bb); k >>= 1
                else:
                    e = x
                    if i & 1:
                        for _ in range(i & 15): e = (mul(e, e) + y) % P
                assert int(out[i]) == e, (field, op, hex(x), hex(y))


def test_ntt_golden_vectors(oracle, libs):
    """vectors from the independent big-int DFT in tests/golden/make_golden.py"""
    import sppark_amd
    for c in json.load(open(os.path.join(HERE, "golden", "ntt_golden.json"))):
        dt = np.uint32 if c["field"] == "bb31" else np.uint64
        x = np.frombuffer(bytes.fromhex(c["input"]), dtype=dt).copy()
        e = np.frombuffer(bytes.fromhex(c["expect"]), dtype=dt)
        sppark_amd.compute_ntt(0, x, c["order"], c["direction"], c["type"], c["field"])
        assert (x == e).all(), c


@pytest.mark.parametrize("field", FIELDS + WIDE)
def test_ntt_vs_oracle_all_modes(oracle, libs, field):
    import sppark_amd
    O = oracle
    f = _oracle_fn(O, field)
    for lg in (list(range(1, 15)) + [16, 18, 20]) if field in FIELDS else (list(range(1, 13)) + [14, 16]):
        x = recipe.ntt_input(field, lg, 7 + lg)
        for order in range(4):
            for direction in range(2):
                for typ in range(2):
                    if lg > 16 and (typ == 1 or direction == 1) and order != 1:
                        continue
                    y = x.copy()
                    sppark_amd.compute_ntt(0, y, order, direction, typ, field)
                    assert (y == f(x, order, direction, typ)).all(), (field, lg, order, direction, typ)


@pytest.mark.parametrize("field", FIELDS + WIDE)
def test_ntt_reference_test_shapes(oracle, libs, field):
    """poc/ntt-cuda/tests/ntt.rs:9-152 and goldilocks_test.go:11-32:
    NTT_NN == NTT_RR; iNTT(NTT(v)) == v in NN and RR; iNTT_RN(NTT_NR(v)) == v;
    coset round trip (the 256-bit fields are additionally pinned against the
    oracle, whose roots are the reference tables', in test_ntt_vs_oracle_all_modes)."""
    import sppark_amd
    from sppark_amd import NTTInputOutputOrder as Ord
    for lg in (list(range(1, 21)) + [22, 24]) if field in FIELDS else (list(range(1, 17)) + [18, 20]):
        v = recipe.ntt_input(field, lg, lg)
        nn = sppark_amd.NTT(0, v.copy(), Ord.NN, field)
        rr = sppark_amd.NTT(0, v.copy(), Ord.RR, field)
        assert (nn == rr).all(), lg
        assert (sppark_amd.iNTT(0, nn.copy(), Ord.NN, field) == v).all(), lg
        assert (sppark_amd.iNTT(0, rr.copy(), Ord.RR, field) == v).all(), lg
        nr = sppark_amd.NTT(0, v.copy(), Ord.NR, field)
        assert (sppark_amd.iNTT(0, nr, Ord.RN, field) == v).all(), lg
        c = sppark_amd.coset_NTT(0, v.copy(), Ord.NN, field)
        assert (sppark_amd.coset_iNTT(0, c, Ord.NN, field) == v).all(), lg


@pytest.mark.parametrize("field", FIELDS)
def test_ntt_linearity_full_size(libs, field):
    """2^24 (BASELINE size): NTT(a) + NTT(b) == NTT(a + b), on device buffers."""
    import torch
    import sppark_amd
    from sppark_amd import NTTInputOutputOrder as Ord
    lg = 24
    p = 0xffffffff00000001 if field == "gl64" else 0x78000001
    a = recipe.ntt_input(field, lg, 1); b = recipe.ntt_input(field, lg, 2)
    if field == "gl64":
        s = a + b
        s = np.where(s < a, s + np.uint64(0xffffffff), s)             # wrapped: + 2^64 mod p
        s = np.where(s >= np.uint64(p), s - np.uint64(p), s)
    else:
        s = ((a.astype(np.uint64) + b) % p).astype(np.uint32)
    def ntt_dev(x):
        t = torch.from_numpy(x.view(np.int64 if field == "gl64" else np.int32)).cuda()
        sppark_amd.NTT(0, t, Ord.NR, field, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return t.cpu().numpy().view(x.dtype)
    A, B, S = ntt_dev(a), ntt_dev(b), ntt_dev(s)
    if field == "gl64":
        t = A + B
        t = np.where(t < A, t + np.uint64(0xffffffff), t)
        t = np.where(t >= np.uint64(p), t - np.uint64(p), t)
    else:
        t = ((A.astype(np.uint64) + B) % p).astype(np.uint32)
    assert (t == S).all()


def _big_input(field, lg, seed):
    """vectorised inputs for sizes where recipe.ntt_input's Python loop (256-bit fields) is too slow:
    four random u64 limbs with the top limb < 2^60, i.e. values < 2^252 < r for both curves"""
    if field in FIELDS:
        return recipe.ntt_input(field, lg, seed)
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 1 << 63, size=(1 << lg, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(1 << lg, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64(0x0fffffffffffffff)
    return x


@pytest.mark.parametrize("field", FIELDS + WIDE)
def test_ntt_full_size_vs_oracle(oracle, libs, field):
    """BASELINE configs[1] / [4] at their own size (2^24 Goldilocks / BabyBear), the WHOLE output
    array against the oracle: forward NR and inverse RN (the timed pair of bench.py), forward NN
    and the coset forward NN.  256-bit fields: 2^22 forward NR + inverse RN, 2^20 NN."""
    import sppark_amd
    O = oracle
    f = _oracle_fn(O, field)
    cases = [(24, O.NR, 0, 0), (24, O.RN, 1, 0), (24, O.NN, 0, 0), (24, O.NN, 0, 1)] if field in FIELDS else \
            [(22, O.NR, 0, 0), (22, O.RN, 1, 0), (20, O.NN, 0, 0)]
    xs = {}
    for lg, order, direction, typ in cases:
        if lg not in xs:
            xs[lg] = _big_input(field, lg, 240 + lg)
        y = xs[lg].copy()
        sppark_amd.compute_ntt(0, y, order, direction, typ, field)
        e = f(xs[lg], order, direction, typ)
        assert y.shape == e.shape and (y == e).all(), (field, lg, order, direction, typ)


@pytest.mark.parametrize("field", FIELDS + WIDE)
def test_ntt_reference_test_range_top(oracle, libs, field):
    """The top of the reference's own test ranges (poc/ntt-cuda/tests/ntt.rs:19,55: lg 1..27 for
    Goldilocks / BabyBear; :114: lg 1..23 for the 256-bit fields), same assertions as the
    reference: NN == RR, iNTT(NTT(v)) == v in NN, RR and NR->RN."""
    import sppark_amd
    from sppark_amd import NTTInputOutputOrder as Ord
    for lg in ((25, 26, 27) if field in FIELDS else (21, 22, 23)):
        v = _big_input(field, lg, 300 + lg)
        nn = sppark_amd.NTT(0, v.copy(), Ord.NN, field)
        rr = sppark_amd.NTT(0, v.copy(), Ord.RR, field)
        assert (nn == rr).all(), lg
        del rr
        assert (sppark_amd.iNTT(0, nn, Ord.NN, field) == v).all(), lg
        nr = sppark_amd.NTT(0, v.copy(), Ord.NR, field)
        assert (sppark_amd.iNTT(0, nr, Ord.RN, field) == v).all(), lg
        rr = sppark_amd.NTT(0, v.copy(), Ord.RR, field)
        assert (sppark_amd.iNTT(0, rr, Ord.RR, field) == v).all(), lg


def test_ntt_lg0_noop_and_errors(libs):
    import sppark_amd
    from sppark_amd import ffi
    x = np.array([5], dtype=np.uint64)
    sppark_amd.NTT(0, x, sppark_amd.NTTInputOutputOrder.NN)
    assert x[0] == 5
    with pytest.raises(ValueError):
        sppark_amd.NTT(0, np.zeros(3, dtype=np.uint64), sppark_amd.NTTInputOutputOrder.NN)
    with pytest.raises(ffi.SpparkError):                     # bad device id -> error, not a crash
        sppark_amd.NTT(99, np.zeros(8, dtype=np.uint64), sppark_amd.NTTInputOutputOrder.NN)


def _field_views(field):
    dt = np.uint32 if field == "bb31" else np.uint64
    w = 4 if field in WIDE else 1
    return dt, w


def test_lde_golden_vectors(oracle, libs):
    """sppark_lde (NTT::LDE_aux) through the C ABI against the committed big-int vectors,
    host buffers and device buffers, with and without the aux output."""
    import torch
    import sppark_amd
    for c in json.load(open(os.path.join(HERE, "golden", "lde_golden.json"))):
        dt, w = _field_views(c["field"])
        x = np.frombuffer(bytes.fromhex(c["input"]), dtype=dt)
        exp = np.frombuffer(bytes.fromhex(c["expect"]), dtype=dt)
        aux_exp = np.frombuffer(bytes.fromhex(c["aux"]), dtype=dt)
        buf = np.zeros(exp.size, dtype=dt); buf[:x.size] = x
        aux = np.zeros(x.size, dtype=dt)
        sppark_amd.LDE(0, buf, c["lg"], c["lg_blowup"], c["field"], aux_out=aux)
        assert (buf == exp).all(), (c["field"], c["lg"], c["lg_blowup"])
        assert (aux == aux_exp).all(), (c["field"], c["lg"], c["lg_blowup"])
        tdt = torch.int32 if dt == np.uint32 else torch.int64
        d = torch.zeros(exp.size, dtype=tdt, device="cuda")
        d[:x.size] = torch.from_numpy(x.view(np.int32 if dt == np.uint32 else np.int64).copy()).cuda()
        sppark_amd.LDE(0, d, c["lg"], c["lg_blowup"], c["field"], stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert (d.cpu().numpy().view(dt) == exp).all(), (c["field"], c["lg"], c["lg_blowup"], "device")


@pytest.mark.parametrize("field,lg,lgb", [("gl64", 10, 2), ("gl64", 15, 1), ("bb31", 12, 3), ("bls12_381", 9, 2), ("bn254", 11, 1)])
def test_lde_vs_oracle(oracle, libs, field, lg, lgb):
    import torch
    import sppark_amd
    O = oracle
    dt, w = _field_views(field)
    x = recipe.ntt_input(field, lg, 77 + lg)
    exp, aux_exp = O.lde(field, x, lgb, want_aux=True)
    buf = np.zeros(((1 << (lg + lgb)), w), dtype=dt); buf[:1 << lg] = x.reshape(-1, w)
    aux = np.zeros((1 << lg, w), dtype=dt)
    sppark_amd.LDE(0, buf, lg, lgb, field, aux_out=aux)
    assert (buf.reshape(exp.shape) == exp).all()
    assert (aux.reshape(aux_exp.shape) == aux_exp).all()
    # LDE_powers / LDE_expand on device buffers
    tdt = torch.int32 if dt == np.uint32 else torch.int64
    sdt = np.int32 if dt == np.uint32 else np.int64
    d_in = torch.from_numpy(np.ascontiguousarray(x).view(sdt).reshape(-1).copy()).cuda()
    d_out = torch.empty(d_in.numel() << lgb, dtype=tdt, device="cuda")
    sppark_amd.LDE_expand(0, d_out, d_in, lg, lgb, field)
    assert (d_out.cpu().numpy().view(dt).reshape(-1, w) == O.lde_expand(field, x, lgb).reshape(-1, w)).all()
    sppark_amd.LDE_powers(0, d_in, field)
    assert (d_in.cpu().numpy().view(dt).reshape(-1, w) == O.lde_powers(field, x).reshape(-1, w)).all()
    # in place: d_in aligned to the END of d_out (ntt/ntt.cuh:358-360)
    buf = torch.full((d_in.numel() << lgb,), 7, dtype=tdt, device="cuda")
    tail = buf[buf.numel() - d_in.numel():]
    tail.copy_(torch.from_numpy(np.ascontiguousarray(x).view(sdt).reshape(-1).copy()).cuda())
    sppark_amd.LDE_expand(0, buf, tail, lg, lgb, field)
    assert (buf.cpu().numpy().view(dt).reshape(-1, w) == O.lde_expand(field, x, lgb).reshape(-1, w)).all()
    from sppark_amd import ffi
    with pytest.raises(ffi.SpparkError):                        # any other overlap is refused
        sppark_amd.LDE_expand(0, d_out, d_out[:d_in.numel()], lg, lgb, field)
    with pytest.raises(ffi.SpparkError):                        # inside d_out, but not at its end
        sppark_amd.LDE_expand(0, d_out, d_out[d_in.numel() // 2:][:d_in.numel()], lg, lgb, field)


def test_lde_large_properties(libs):
    """2^20 -> 2^22 Goldilocks: the extension restricted to every 4th coset point is the coset
    NTT of the original polynomial, and iNTT(coset) of the whole extension returns the padded
    coefficients -- size-independent checks at a size the oracle is too slow for."""
    import torch
    import sppark_amd
    from sppark_amd import NTTInputOutputOrder as Ord
    lg, lgb = 20, 2
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = (torch.randint(0, 2**62, (1 << lg,), dtype=torch.int64, device="cuda", generator=g))
    ext = torch.zeros(1 << (lg + lgb), dtype=torch.int64, device="cuda"); ext[:1 << lg] = x
    aux = torch.zeros(1 << lg, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    sppark_amd.LDE(0, ext, lg, lgb, "gl64", aux_out=aux, stream=s)
    coef = x.clone(); sppark_amd.iNTT(0, coef, Ord.NN, "gl64", stream=s)
    torch.cuda.synchronize()
    assert torch.equal(coef, aux)
    back = ext.clone(); sppark_amd.coset_iNTT(0, back, Ord.NN, "gl64", stream=s)
    torch.cuda.synchronize()
    assert torch.equal(back[:1 << lg], coef) and int(back[1 << lg:].abs().sum()) == 0


def test_lde_and_msm_argument_errors(libs):
    """Invalid arguments come back as a non-zero {code, message} (no crash, no C++ exception across the
    boundary): domain larger than the field's two-adicity, NULL inputs, host pointers where device
    pointers are required."""
    import torch
    import sppark_amd
    from sppark_amd import ffi
    L = ffi.load("bb31")                                        # two-adicity 27
    d = torch.zeros(16, dtype=torch.int32, device="cuda")
    for call in (lambda: L.sppark_lde(0, d.data_ptr(), 26, 2, None, None),
                 lambda: L.sppark_lde_powers(0, d.data_ptr(), 28, None),
                 lambda: L.sppark_lde_expand(0, d.data_ptr(), d.data_ptr(), 2, 1, None),          # overlapping
                 lambda: L.sppark_lde_powers(0, np.zeros(16, dtype=np.uint32).ctypes.data, 4, None)):   # host pointer
        err = call()
        assert err.code != 0
        with pytest.raises(ffi.SpparkError):
            ffi.check(L, err)
    M = ffi.load("bls12_381")
    out = np.ones(144, dtype=np.uint8)
    err = M.mult_pippenger_inf(out.ctypes.data, None, 5, np.zeros(160, dtype=np.uint8).ctypes.data, 104)
    assert err.code != 0 and (out == 0).all()                   # out = infinity on error (pippenger.cuh:740)
    with pytest.raises(ffi.SpparkError):
        ffi.check(M, err)


@pytest.mark.parametrize("field", FIELDS + WIDE)
def test_lde_sweep_vs_oracle(oracle, libs, field):
    """every (domain, blow-up) up to 2^12 x 8 against the oracle's restatement of NTT::LDE_aux,
    host buffers, with the coefficient output."""
    import sppark_amd
    O = oracle
    dt, w = _field_views(field)
    top = 12 if field in FIELDS else 9
    for lg in range(0, top + 1):
        for lgb in (1, 2, 3):
            if field == "bb31" and lg + lgb > 27:
                continue
            x = recipe.ntt_input(field, lg, 900 + 17 * lg + lgb)
            exp, aux_exp = O.lde(field, x, lgb, want_aux=True)
            buf = np.zeros(((1 << (lg + lgb)), w), dtype=dt); buf[:1 << lg] = x.reshape(-1, w)
            aux = np.zeros((1 << lg, w), dtype=dt)
            sppark_amd.LDE(0, buf, lg, lgb, field, aux_out=aux)
            assert (buf.reshape(exp.shape) == exp).all(), (field, lg, lgb)
            assert (aux.reshape(aux_exp.shape) == aux_exp).all(), (field, lg, lgb)


@pytest.mark.parametrize("lib,field", [("gl64_plonky2", "gl64"), ("bb31_canonical", "bb31")])
def test_ntt_root_convention_variants(oracle, libs, lib, field):
    """libsppark_gl64_plonky2.so (-DGOLDILOCKS_PLONKY2, ntt/parameters/goldilocks.h:7-82) and
    libsppark_bb31_canonical.so (-DBABY_BEAR_CANONICAL, baby_bear.h:7-74): every order / direction /
    type against the oracle switched to the same convention (pinned against the reference's table
    entries in tests/test_oracle.py), and the LDE (uses the variant's coset generator)."""
    import sppark_amd
    O = oracle
    f = _oracle_fn(O, field)
    try:
        O.set_root_conventions(True, True)
        for lg in list(range(1, 13)) + [16, 20]:
            x = recipe.ntt_input(field, lg, 70 + lg)
            for order in range(4):
                for direction in range(2):
                    for typ in range(2):
                        if lg > 12 and (typ == 1 or direction == 1) and order != 1:
                            continue
                        y = x.copy()
                        sppark_amd.compute_ntt(0, y, order, direction, typ, lib)
                        assert (y == f(x, order, direction, typ)).all(), (lib, lg, order, direction, typ)
        x = recipe.ntt_input(field, 10, 5)
        ext = np.zeros(1 << 12, dtype=x.dtype); ext[:1 << 10] = x
        sppark_amd.LDE(0, ext, 10, 2, lib)
        assert (ext == O.lde(field, x, 2)).all()
        # and the default libraries do NOT use these roots
        y = x.copy(); sppark_amd.compute_ntt(0, y, 0, 0, 0, field)
        assert not (y == f(x, 0, 0, 0)).all()
    finally:
        O.set_root_conventions(False, False)


@pytest.mark.parametrize("field", FIELDS)
def test_ntt_plan_knobs_in_a_fresh_process(oracle, libs, field):
    """The plan knobs are read once per process: the 8-stage plan (SPPARK_NTT_R64_MIN=99, what the single-word fields ran
    until round 3 and the 256-bit fields still run) and the radix-64 plan with two small inter-pass tables everywhere
    (SPPARK_NTT_R64_DIRECT=12) against the oracle, each in its own interpreter."""
    import subprocess, sys
    code = (
        "import sys, os, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests', 'golden'))\n"
        "import oracle as O, recipe, sppark_amd\n"
        "f = O.ntt_gl64 if %r == 'gl64' else O.ntt_bb31\n"
        "for lg in (12, 13, 16, 18, 20):\n"
        "    x = recipe.ntt_input(%r, lg, 40 + lg)\n"
        "    for order, direction in ((1, 0), (2, 1), (0, 0), (3, 1)):\n"
        "        y = x.copy(); sppark_amd.compute_ntt(0, y, order, direction, 0, %r)\n"
        "        assert (y == f(x, order, direction, 0)).all(), (lg, order, direction)\n"
        "print('ok')\n") % (os.path.dirname(HERE), os.path.dirname(HERE), field, field, field)
    for env in ({"SPPARK_NTT_R64_MIN": "99"}, {"SPPARK_NTT_R64_DIRECT": "12"}, {"SPPARK_NTT_R64_DIRECT": "24"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stderr[-2000:])


def test_ntt_cached_tables_and_scratch_are_bounded_and_releasable(oracle, libs):
    """The caches behind compute_ntt (round 4, the advisor's findings): the inter-pass twiddle tables are shared by
    all transform sizes (a second size adds only its own small tables), sppark_ntt_release_cached frees tables and
    idle scratch, and everything is rebuilt transparently on the next call; an unaligned device view takes the
    element-wise bit reversal."""
    import torch
    import sppark_amd
    from sppark_amd import ffi
    O = oracle
    Ord = sppark_amd.NTTInputOutputOrder
    for field in ("bls12_381", "gl64", "bb31"):
        L = ffi.load(field)
        f = _oracle_fn(O, field)
        L.sppark_ntt_release_cached()
        assert L.sppark_ntt_cached_tables() == 0 and L.sppark_ntt_cached_scratch_bytes() == 0
        x = recipe.ntt_input(field, 16, 5)
        y = x.copy()
        sppark_amd.NTT(0, y, Ord.NR, field)                     # host buffer: stages through the pool
        assert (y == f(x, 1, 0, 0)).all()
        t16 = L.sppark_ntt_cached_tables()
        assert t16 > 1 and 0 < L.sppark_ntt_cached_scratch_bytes() <= 1 << 30
        # a smaller transform that is the lower passes of the 2^16 one (radix-64 plan: 2^12; the 256-bit fields' passes
        # of 8 stages: 2^8) adds its own root tables and finds the inter-pass tables -- keyed by the sub-problem size,
        # not by the transform size -- already there
        x2 = recipe.ntt_input(field, 8 if field in WIDE else 12, 6)
        y2 = x2.copy()
        sppark_amd.NTT(0, y2, Ord.NR, field)
        assert (y2 == f(x2, 1, 0, 0)).all()
        assert L.sppark_ntt_cached_tables() == t16 + 1
        z = x.copy()
        sppark_amd.iNTT(0, z, Ord.RN, field)                    # the table that carries 1/n belongs to this size
        assert (z == f(x, 2, 1, 0)).all()
        L.sppark_ntt_release_cached()
        assert L.sppark_ntt_cached_tables() == 0 and L.sppark_ntt_cached_scratch_bytes() == 0
        y3 = x.copy()
        sppark_amd.NTT(0, y3, Ord.NN, field)
        assert (y3 == f(x, 0, 0, 0)).all()
    # a device view at an odd element offset (4-byte aligned only): NN goes through the element-wise tiles
    x = recipe.ntt_input("bb31", 14, 9)
    buf = torch.zeros(x.size + 1, dtype=torch.int32, device="cuda")
    buf[1:] = torch.from_numpy(x.view(np.int32)).cuda()
    sppark_amd.NTT(0, buf[1:], Ord.NN, "bb31", stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (buf[1:].cpu().numpy().view(np.uint32) == O.ntt_bb31(x, 0, 0, 0)).all()
