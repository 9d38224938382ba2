"""The subset-sum top cut into pieces (bucket_top_piece): pieces per subset sum / per plain sum, swept at the sizes whose top
has 2048 / 4096 items.  Needs a TUNING build (SPPARK_LIBDIR=lib_tuning, -DSPPARK_TUNING: SPPARK_TOP_CUT is read per MSM).
Usage: SPPARK_LIBDIR=lib_tuning python tools/gpu_msm_top_cut.py 18 20 21 22"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sppark_amd                       # noqa: E402
from sppark_amd import synth            # noqa: E402

curve = "bls12_381"
for lg in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << lg
    pts, _ = synth.replicated_points(n, curve)
    sc = synth.uniform_scalars(n, curve, seed=5)
    ctx = sppark_amd.MsmContext(curve, stream=torch.cuda.current_stream().cuda_stream)
    ctx.enable_timing(True)
    ref = None
    for cut in ("", "11", "12", "14", "22", "24", "28", "", "24"):
        if cut: os.environ["SPPARK_TOP_CUT"] = cut
        else: os.environ.pop("SPPARK_TOP_CUT", None)
        best = None
        for rep in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = ctx.invoke(pts, sc)
            wall = (time.perf_counter() - t0) * 1e3
            d, a, b = ctx.kernel_ms(2), ctx.kernel_ms(1), ctx.kernel_ms(0)
            if rep >= 2 and (best is None or d < best[0]): best = (d, a, b, wall)
        aff = sppark_amd.to_affine(out, curve)
        if ref is None: ref = aff
        assert (aff == ref).all(), cut
        d, a, b, wall = best
        print("2^%d cut %-4s: tail %.3f device %.3f wall %.3f" % (lg, cut or "auto", d - a - b, d, wall), flush=True)
    ctx.close()
