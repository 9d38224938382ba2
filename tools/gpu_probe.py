"""GPU probe: instruction micro-benchmarks, field-level throughput and first MSM
timings.  Writes a JSON summary to gpurun_out/."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sppark_amd
from sppark_amd import ffi

OUT = {}
NAMES = ["mad_u64_u32 x8 indep", "mad_u64_u32 x8 dep", "mad+addc x8 dep", "mad+addc 2 chains x8", "mad+addc+s_nop x8 dep",
         "mul_lo_u32 x8 dep", "mul_hi_u32 x8 dep", "mad_u32_u24 x8 dep", "add_co+addc x8 dep", "fma_f64 x8 (4 chains)",
         "lshl_add_u64 x8 dep", "add_u32 x8 dep", "mul_lo_u32 x8 (4 chains)"]

print("== MSM timing (device-resident inputs) ==")
msmres = {}
for curve, fb in (("bls12_381", 48), ("bn254", 32)):
    base = torch.zeros((2048, 2 * fb), dtype=torch.uint8, device="cuda")
    sppark_amd.generate_points(base, 2048, 0x5eed5eed0001, 2 * fb, curve)
    ctx = sppark_amd.MsmContext(curve)
    ctx.enable_timing(True)
    for lg in (16, 20, 22, 24, 26):
        if len(sys.argv) > 1 and str(lg) not in sys.argv[1:]:
            continue
        if curve == "bn254" and lg not in (20, 24, 26):
            continue
        n = 1 << lg
        idx = torch.arange(n, device="cuda") % 2048
        pts = base[idx].contiguous()
        g = torch.Generator(device="cuda"); g.manual_seed(lg)
        sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        sc[:, 31] &= 0x1f
        torch.cuda.synchronize()
        for wb in ((0,) if lg < 24 else (0, 16, 18, 20, 22)):
            ctx.tune(wbits=wb)
            t0 = time.time(); out = ctx.invoke(pts, sc); t1 = time.time()       # includes allocation
            t0 = time.time(); out = ctx.invoke(pts, sc); torch.cuda.synchronize(); t1 = time.time()
            r = {"wall_ms": (t1 - t0) * 1e3, "sort_ms": ctx.kernel_ms(0), "accum_ms": ctx.kernel_ms(1),
                 "device_ms": ctx.kernel_ms(2), "scratch_GB": ctx.scratch_bytes() / 1e9,
                 "pts_per_s": n / (t1 - t0)}
            msmres["%s 2^%d wbits=%d" % (curve, lg, wb)] = r
            print("%-10s 2^%d wbits=%-2d  wall %9.2f ms  sort %8.2f  accum %9.2f  device %9.2f  scratch %.2f GB  %.3e pts/s"
                  % (curve, lg, wb, r["wall_ms"], r["sort_ms"], r["accum_ms"], r["device_ms"], r["scratch_GB"], r["pts_per_s"]))
            OUT["msm"] = msmres
            json.dump(OUT, open("gpurun_out/probe_msm.json", "w"), indent=1)
        del pts, sc
    ctx.close()
