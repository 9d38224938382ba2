"""Level-A sort records: 4 bytes (the automatic plan) against 8 bytes (an explicit slab count keeps the wide records,
msm_plan.hpp; the slab counts asked for are the automatic ones), alternating on one box.  Per size: the median of REPS
calls of the time before the first accumulation (digits + sort + point conversion) and of the whole device part.

    python tools/gpu_msm_records_ab.py [lg | lg+extra ...]            (default 26 24 22 20; "26+4096" = 2^26 + 4096 points)
"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
import oracle as O                   # (Jacobian -> affine for the comparison of the two results only)
REPS = 7
sizes = [(lambda t: (1 << int(t[0])) + (int(t[1]) if len(t) > 1 else 0))(a.split("+")) for a in sys.argv[1:]] or [1 << 26, 1 << 24, 1 << 22, 1 << 20]
base = torch.zeros((2048, 96), dtype=torch.uint8, device="cuda")
sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 96)
for n in sizes:
    lg = n.bit_length() - 1
    pts = base[torch.arange(n, device="cuda") % 2048].contiguous()
    g = torch.Generator(device="cuda"); g.manual_seed(lg)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3f
    ctx = sppark_amd.MsmContext("bls12_381"); ctx.enable_timing(True)
    nslabs = min(64, max(n // 131072, min(8, max(1, n // 2048))))       # msm_plan.hpp: the automatic slab count
    res = {"packed": ([], []), "wide": ([], [])}
    ref = None
    for rep in range(REPS + 1):
        for name, ns in (("packed", 0), ("wide", nslabs)):
            ctx.tune(nslabs=ns)
            out = O.jac_to_affine(0, ctx.invoke(pts, sc))     # (the order inside a bucket, hence the Jacobian triple, is free)
            if ref is None:
                ref = out.copy()
            assert (out == ref).all(), (lg, name)
            if rep:
                res[name][0].append(ctx.kernel_ms(0)); res[name][1].append(ctx.kernel_ms(2))
    m = {k: (statistics.median(v[0]), statistics.median(v[1])) for k, v in res.items()}
    print("%s: before the accumulation %.2f ms with 8-byte records -> %.2f with 4-byte ones; device part %.2f -> %.2f ms"
          % ("2^%d" % lg if n == 1 << lg else "2^%d + %d" % (lg, n - (1 << lg)), m["wide"][0], m["packed"][0], m["wide"][1], m["packed"][1]), flush=True)
    del ctx, pts, sc
