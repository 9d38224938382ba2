"""Tile-geometry sweep for the NTT passes (env knobs read by ntt_driver.hpp on every call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sppark_amd
from sppark_amd import NTTInputOutputOrder as Ord

torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
stream = torch.cuda.current_stream().cuda_stream
fields = sys.argv[1:] or ["gl64", "bb31"]
for field in fields:
    dt, eb = (torch.int64, 8) if field == "gl64" else (torch.int32, 4)
    base = 4 if eb == 8 else 5
    for lg in (20, 24, 26):
        n = 1 << lg
        x = torch.randint(0, 2**30, (n,), dtype=dt, device="cuda")
        for smax in (8, 7, 6):
            for lgc in (base, base + 1, base + 2):
                for lgt in (base + 8, base + 9):
                    os.environ["SPPARK_NTT_SMAX"] = str(smax)
                    os.environ["SPPARK_NTT_LGC"] = str(lgc)
                    os.environ["SPPARK_NTT_LGTILE"] = str(lgt)
                    try:
                        for _ in range(3):
                            sppark_amd.NTT(0, x, Ord.NR, field, stream=stream)
                        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                        reps = 20
                        e0.record()
                        for _ in range(reps):
                            sppark_amd.NTT(0, x, Ord.NR, field, stream=stream)
                        e1.record(); torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1) / reps
                        print("%s 2^%d smax=%d lgC=%d lgtile=%d: %.3f ms  %.2e el/s" % (field, lg, smax, lgc, lgt, ms, n / ms * 1e3), flush=True)
                    except Exception as ex:
                        print("%s 2^%d smax=%d lgC=%d lgtile=%d: FAILED %s" % (field, lg, smax, lgc, lgt, str(ex)[:80]), flush=True)
