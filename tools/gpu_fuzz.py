"""Time-bounded randomised parity run on the GPU, beyond the seeded cases of tests/: every result against the oracle
(the checker; never the thing measured).  Usage: python tools/gpu_fuzz.py SECONDS [SEED] [mid]

MSM: all G1 curves, G2 of the curves that have it; sizes 1 .. 2^18 (log-uniform, ragged), few / many distinct points,
flagged and plain layouts, infinities, P / -P pairs, repeated (point, scalar) pairs; scalar shapes: uniform, all equal,
half zero, short, r - 1 / (r +- 1) / 2 heavy, digits clustered in one bucket; random plan tunables (window bits, run
length, fan-in, chunks, slabs, sort split, top hand-over, tail variant, record format).
mid: MSMs of 2^17 .. 2^24.3 points (ragged sizes, where the plan changes run length, slabs, index groups of the 4-byte
sort records, oversized partitions, window groups) over a window of the ALL-DISTINCT progression P_i = (a + i b) G
(sppark_g1_generate_progression), device-resident, expected (a sum s_i + b sum i s_i mod r) G from integer arithmetic on the
scalars (oracle/fold.py) and one oracle scalar multiplication: any wrong gather index changes the result.
api: the other ways into the same path at 1 .. 2^16 points: host / device buffers mixed, window groups and chunks that do not
divide n, a bounded scratch, preloaded bases (a prefix of them), fixed-base tables with forced widths, bitmap batch
additions (with and without a reference map), LDEs of every field (2^0 .. 2^12, blow-up 2 .. 8, with the coefficient output).
NTT: all fields, sizes 2^1 .. 2^20, the 16 modes, inputs heavy in the values where a reduction can go wrong
(0, 1, p - 1, p - 2, 2^32 - 1, 2^32, 2^64 - 2^32 ..., all-equal arrays, one-hot arrays)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import oracle as O          # noqa: E402  (checker)
import recipe               # noqa: E402
import sppark_amd           # noqa: E402

G1 = [(O.CURVE_ID[name], name) for name in ("bls12_381", "bn254", "bls12_377", "pallas", "vesta")]
G2 = ["bls12_381", "bn254", "bls12_377"]
NTT_FIELDS = ["gl64", "bb31", "bls12_381", "bn254", "bls12_377", "pallas", "vesta"]


def scalar_shape(rng, curve, sc, n, mode):
    r = O.FR_MODULUS[curve]
    le = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
    if mode == 1: sc[:] = sc[0]
    elif mode == 2: sc[rng.integers(0, 2, size=n).astype(bool)] = 0
    elif mode == 3: sc[:, int(rng.integers(1, 12)):] = 0
    elif mode == 4 and n > 4: sc[: n // 2] = sc[n // 2: 2 * (n // 2)]
    elif mode == 5:
        special = [le(r - 1), le((r + 1) // 2), le((r - 1) // 2), le(1), le(2), le(r - 2)]
        pick = rng.integers(0, 2 * len(special), size=n)
        for k, s in enumerate(special):
            sc[pick == k] = s
    elif mode == 6:                                     # every scalar differs from one value in its low 16 bits only
        sc[:, 2:] = sc[0, 2:]
    elif mode == 7:                                     # one heavy bucket in every window, a few stragglers
        keep = rng.integers(0, 16, size=n) == 0
        sc[~keep] = sc[0]
    return sc


def fuzz_msm(rng, it, ctxs, g2):
    if g2:
        name = G2[it % len(G2)]; curve = O.CURVE_ID_G2[name]
    else:
        curve, name = G1[it % len(G1)]
    lg = rng.uniform(0, 14 if g2 else 18)
    n = max(1, int(2 ** lg) + int(rng.integers(-3, 4)))
    nd = int(rng.choice([1, 2, 7, 64, 509, 2048]))
    flagged = bool(rng.integers(0, 2))
    pts, sc = recipe.msm_inputs(curve, n, int(rng.integers(1, 1 << 30)), ndistinct=nd, flagged=flagged, edge=bool(rng.integers(0, 2)))
    mode = int(rng.integers(0, 8))
    sc = scalar_shape(rng, O.CURVE_ID[name], sc, n, mode)
    what = dict(kind="g2" if g2 else "g1", curve=name, n=n, nd=nd, flagged=flagged, mode=mode)
    if g2:
        out = sppark_amd.multi_scalar_mult_fp2_arkworks(pts, sc, name, ffi_affine_sz=pts.shape[1])
        exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
        got = sppark_amd.to_affine_g2(out, name)
    else:
        ctx = ctxs[name]
        t = dict(wbits=int(rng.choice([0, 0, 0, 3, 5, 8, 11, 14, 17, 20])), L=int(rng.choice([0, 0, 4, 16, 35, 64, 256])),
                 F=int(rng.choice([0, 4, 8, 32])), K=int(rng.choice([0, 2, 8])), nslabs=int(rng.choice([0, 0, 1, 3, 8])))
        ctx.tune(**t)
        ts = int(rng.choice([0, 0, 1, 4])); ctx.tune_sort(ts)
        top = int(rng.choice([0, 0, 1, 64, 4096])); ctx.tune_sums(top)
        join = int(rng.choice([0, 0, 0, 1, 5, 6, 8, 26, 29, 46])); k1 = int(rng.choice([0, 0, 2, 8])); ctx.tune_tail(join, k1)
        rec = int(rng.choice([0, 0, 1, 2])); ctx.tune_records(rec)
        what.update(t, sort=ts, top=top, join=join, k1=k1, records=rec)
        out = ctx.invoke(pts, sc, ffi_affine_sz=pts.shape[1])
        exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
        got = sppark_amd.to_affine(out, name)
    return bool((got == exp).all()), what


def fuzz_api(rng, it, state):
    import torch
    sel = it % 4
    if sel == 3:                                                    # LDE
        field = NTT_FIELDS[(it // 4) % len(NTT_FIELDS)]
        small = field in ("gl64", "bb31")
        lg, lgb = int(rng.integers(0, 13 if small else 10)), int(rng.integers(1, 4))
        x, _ = ntt_edge_input(rng, field, lg) if lg else (recipe.ntt_input(field, 0, 5), 0)
        w = 1 if small else 4
        exp, aux_exp = O.lde(field, x, lgb, want_aux=True)
        buf = np.zeros(((1 << (lg + lgb)), w), dtype=x.dtype); buf[:1 << lg] = x.reshape(-1, w)
        aux = np.zeros((1 << lg, w), dtype=x.dtype)
        sppark_amd.LDE(0, buf, lg, lgb, field, aux_out=aux)
        return bool((buf.reshape(exp.shape) == exp).all() and (aux.reshape(aux_exp.shape) == aux_exp).all()), dict(kind="api", what="lde", field=field, lg=lg, lgb=lgb)
    curve, name = G1[(it // 4) % len(G1)]
    n = max(1, int(2 ** rng.uniform(0, 16)) + int(rng.integers(-3, 4)))
    flagged = bool(rng.integers(0, 2))
    pts, sc = recipe.msm_inputs(curve, n, int(rng.integers(1, 1 << 30)), ndistinct=int(rng.choice([1, 3, 64, 700])), flagged=flagged, edge=bool(rng.integers(0, 2)))
    sc = scalar_shape(rng, curve, sc, n, int(rng.integers(0, 8)))
    st = pts.shape[1]
    dev = lambda a: torch.from_numpy(a).cuda()
    if sel == 2:                                                    # bitmap batch addition
        r = O.FR_MODULUS[curve]
        words = (n + 31) // 32
        bm = rng.random(words * 32) < rng.choice([0.0, 0.03, 0.5, 1.0]); rm = rng.random(words * 32) < 0.3
        use_ref = bool(rng.integers(0, 2))
        s = np.zeros((n, 32), dtype=np.uint8)
        pick = bm ^ rm if use_ref else bm
        s[pick[:n]] = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
        if use_ref: s[(pick & rm)[:n]] = np.frombuffer((r - 1).to_bytes(32, "little"), dtype=np.uint8)
        bw = np.packbits(bm.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
        rw = np.packbits(rm.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1)
        out = sppark_amd.batch_addition(pts, bw, rw if use_ref else None, name, ffi_affine_sz=st)
        return bool((sppark_amd.to_affine(out, name) == O.msm_affine(curve, pts, s, algo=0, param=4)).all()), dict(kind="api", what="batch_addition", curve=name, n=n, ref=use_ref)
    ctx = sppark_amd.MsmContext(name)
    try:
        exp = O.msm_affine(curve, pts, sc, algo=0, param=8)
        if sel == 0:                                                # buffers x groups x chunks x scratch bound
            groups = int(rng.choice([0, 1, 2, 3, 5])); chunk = int(rng.choice([0, 0, 1 + n // 3, 4096, 7001]))
            ctx.tune(wbits=int(rng.choice([0, 0, 9, 13])))
            ctx.tune_pipeline(groups=groups, chunk_points=chunk, max_scratch_bytes=int(rng.choice([0, 0, 0, 1 << 26])))
            hp, hs = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            out = ctx.invoke(pts if hp else dev(pts), sc if hs else dev(sc), ffi_affine_sz=st)
            what = dict(what="pipeline", groups=groups, chunk=chunk, host_points=hp, host_scalars=hs)
        else:                                                       # preloaded bases, plain or with fixed-base tables
            fixed = bool(rng.integers(0, 2)); wb = int(rng.choice([0, 8, 10, 13, 16, 21])) if fixed else 0
            ctx.tune(wbits=wb)
            ctx.set_points(pts if rng.integers(0, 2) else dev(pts), ffi_affine_sz=st, fixed_base=fixed)
            ctx.tune(wbits=0)
            m = n if rng.integers(0, 2) else max(1, n // int(rng.integers(2, 5)))
            s_ = np.ascontiguousarray(sc[:m])
            out = ctx.invoke(None, s_ if rng.integers(0, 2) else dev(s_), npoints=m)
            if m < n: exp = O.msm_affine(curve, pts[:m], s_, algo=0, param=8)
            what = dict(what="preloaded", fixed=fixed, wbits=wb, m=m)
        return bool((sppark_amd.to_affine(out, name) == exp).all()), dict(kind="api", curve=name, n=n, flagged=flagged, **what)
    finally:
        ctx.close()


A_PROG, B_PROG = 0x243f6a8885a308d313198a2e03707344, 0xa4093822299f31d0082efa99       # < 2^126, < 2^96
MID_MAX = (1 << 24) + (1 << 22)


def fuzz_mid(rng, it, state):
    import torch
    from sppark_amd import synth
    from oracle import fold
    name = ("bls12_381", "bn254")[it % 2]; curve = O.CURVE_ID[name]
    fb = O.FP_BYTES[curve]; r = O.FR_MODULUS[curve]
    if name not in state:
        pts = torch.empty((MID_MAX, 2 * fb), dtype=torch.uint8, device="cuda")
        sppark_amd.generate_progression(pts, MID_MAX, A_PROG, B_PROG, 2 * fb, name)
        state[name] = (pts, sppark_amd.MsmContext(name, stream=torch.cuda.current_stream().cuda_stream), O.g1_generator(curve))
    pts, ctx, G = state[name]
    n = min(MID_MAX, int(2 ** rng.uniform(17, 24.3)) + int(rng.integers(-5, 6)))
    o = int(rng.integers(0, MID_MAX - n + 1))
    sc = synth.uniform_scalars(n, name, seed=int(rng.integers(1, 1 << 30)))
    mode = int(rng.integers(0, 6))
    if mode == 1: sc[:] = sc[0].clone()
    elif mode == 2: sc[torch.rand(n, device="cuda") < 0.5] = 0
    elif mode == 3: sc[:, int(rng.integers(2, 20)):] = 0
    elif mode == 4: sc[torch.rand(n, device="cuda") < 0.97] = sc[0].clone()
    elif mode == 5: sc[int(rng.integers(0, n // 2)): n - int(rng.integers(0, n // 2))] = sc[1].clone()
    t = dict(wbits=int(rng.choice([0, 0, 0, 12, 15, 18, 21, 23])), L=int(rng.choice([0, 0, 0, 32, 100, 128, 256])),
             F=int(rng.choice([0, 0, 4, 32])), K=int(rng.choice([0, 0, 2, 8])), nslabs=int(rng.choice([0, 0, 0, 5, 64])))
    ctx.tune(**t)
    ts = int(rng.choice([0, 0, 13, 10])); ctx.tune_sort(ts)
    big = int(rng.choice([0, 0, 30000])); ctx.tune_split(big)
    top = int(rng.choice([0, 0, 1, 512, 4096])); ctx.tune_sums(top)
    join = int(rng.choice([0, 0, 1, 6])); k1 = int(rng.choice([0, 0, 4, 16])); ctx.tune_tail(join, k1)
    rec = int(rng.choice([0, 0, 1, 2])); ctx.tune_records(rec)
    groups = int(rng.choice([0, 0, 2, 3])); ctx.tune_pipeline(groups=groups)
    what = dict(kind="mid", curve=name, n=n, offset=o, mode=mode, sort=ts, big=big, top=top, join=join, k1=k1, records=rec, groups=groups, **t)
    torch.cuda.synchronize()
    got = sppark_amd.to_affine(ctx.invoke(pts[o:o + n], sc), name)
    s0, s1 = fold.weighted_sums(sc)
    exp = O.g1_mul(curve, G, ((A_PROG + o * B_PROG) * s0 + B_PROG * s1) % r)
    return bool((got == exp).all()), what


def ntt_edge_input(rng, field, lg):
    n = 1 << lg
    x = recipe.ntt_input(field, lg, int(rng.integers(1, 1 << 30)))
    shape = int(rng.integers(0, 5))
    if field in ("gl64", "bb31"):
        p = O.GL64_P if field == "gl64" else O.BB31_P
        if field == "gl64":
            edges = [0, 1, p - 1, p - 2, 0xffffffff, 1 << 32, (1 << 32) + 1, 0xffffffff00000000, 0xfffffffeffffffff, 1 << 63, (1 << 63) - 1]
        else:
            edges = [0, 1, p - 1, p - 2, 0x7fffffff % p, 1 << 27, (1 << 27) - 1, 0x77ffffff]
        e = np.array(edges, dtype=x.dtype)
        if shape == 1: x[:] = e[rng.integers(0, len(e), size=n)]
        elif shape == 2:
            m = rng.integers(0, 2, size=n).astype(bool); x[m] = e[rng.integers(0, len(e), size=int(m.sum()))]
        elif shape == 3: x[:] = e[int(rng.integers(0, len(e)))]
        elif shape == 4: x[:] = 0; x[int(rng.integers(0, n))] = e[int(rng.integers(1, len(e)))]
    else:
        r = O.FR_MODULUS[O.CURVE_ID[field]]
        limbs = lambda v: np.array([(v >> (64 * k)) & 0xffffffffffffffff for k in range(4)], dtype=np.uint64)
        edges = [limbs(v) for v in (0, 1, r - 1, r - 2, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, (1 << 192), (r - 1) // 2, (1 << 254) % r)]
        if shape == 1:
            pick = rng.integers(0, len(edges), size=n)
            for k, e in enumerate(edges): x[pick == k] = e
        elif shape == 2:
            pick = rng.integers(0, 2 * len(edges), size=n)
            for k, e in enumerate(edges): x[pick == k] = e
        elif shape == 3: x[:] = edges[int(rng.integers(0, len(edges)))]
        elif shape == 4: x[:] = 0; x[int(rng.integers(0, n))] = edges[int(rng.integers(1, len(edges)))]
    return x, shape


def fuzz_ntt(rng, it):
    field = NTT_FIELDS[it % len(NTT_FIELDS)]
    small = field in ("gl64", "bb31")
    lg = int(rng.integers(1, 21 if small else 15))
    x, shape = ntt_edge_input(rng, field, lg)
    order, direction, typ = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
    if small:
        f = O.ntt_gl64 if field == "gl64" else O.ntt_bb31
        exp = f(x, order, direction, typ)
    else:
        exp = O.ntt_fr(O.CURVE_ID[field], x, order, direction, typ)
    y = x.copy()
    sppark_amd.compute_ntt(0, y, order, direction, typ, field)
    return bool((y == exp).all()), dict(kind="ntt", field=field, lg=lg, order=order, direction=direction, type=typ, shape=shape)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    mid = len(sys.argv) > 3 and sys.argv[3] == "mid"
    api = len(sys.argv) > 3 and sys.argv[3] == "api"
    ctxs = {} if mid or api else {name: sppark_amd.MsmContext(name) for _, name in G1}
    state = {}
    t0 = time.time()
    counts = {"g1": 0, "g2": 0, "ntt": 0, "mid": 0, "api": 0}
    points = 0
    bad = []
    it = 0
    while time.time() - t0 < budget:
        sel = it % 8
        try:
            if mid: ok, what = fuzz_mid(rng, it, state)
            elif api: ok, what = fuzz_api(rng, it, state)
            elif sel < 4: ok, what = fuzz_msm(rng, it // 8 * 4 + sel, ctxs, False)
            elif sel == 4: ok, what = fuzz_msm(rng, it // 8, ctxs, True)
            else: ok, what = fuzz_ntt(rng, it // 8 * 3 + sel - 5)
        except Exception as e:                                      # an error return is a finding too: report, go on
            ok, what = False, dict(kind=("mid" if mid else "api" if api else "g1" if sel < 4 else "g2" if sel == 4 else "ntt"), iteration=it, error=repr(e))
        counts[what["kind"]] += 1
        points += what.get("n", 0)
        if not ok:
            bad.append(what); print("MISMATCH", what, flush=True)
        it += 1
    redo = {name: c.tail_redone() for name, c in ctxs.items()}
    for c in ctxs.values():
        c.close()
    if mid or api:
        for _, c, _ in state.values():
            c.close()
        print("seed %d, %.0f s: %d %s; mismatches: %d"
              % (seed, time.time() - t0, counts["mid"] + counts["api"],
                 "mid-size MSMs over the all-distinct progression (%d points in all)" % points if mid else
                 "calls through the pipeline / preloaded / fixed-base / batch-addition / LDE entry points", len(bad)))
        sys.exit(1 if bad else 0)
    print("seed %d, %.0f s: %d G1 MSMs (%d points in all), %d G2 MSMs, %d NTTs against the oracle; mismatches: %d; tails redone: %s"
          % (seed, time.time() - t0, counts["g1"], points, counts["g2"], counts["ntt"], len(bad), redo))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
