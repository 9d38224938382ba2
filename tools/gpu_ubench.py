"""Instruction micro-benchmarks on gfx950 (see csrc/api/devtest_api.hip k_ub<>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sppark_amd import ffi
L = ffi.load("bn254")
NAMES = ["mad_u64_u32 x8 indep", "mad_u64_u32 x8 dep", "mad+addc x8 dep", "mad+addc 4 chains x2", "mul_lo_u32 x8 dep",
         "mul_lo_u32 x8 indep", "mad_u32_u24 x8 indep", "add_u32 x8 indep", "add_u32 x8 dep", "fma_f64 x8 indep",
         "fma_f64 x8 dep", "lshl_add_u64 x8 indep", "add_co+addc x4 pairs indep", "mul_hi_u32 x8 indep"]
print("%-28s %s" % ("block of 8 instr", "waves/SIMD: cycles per block (one wave's view) | wave-instr/clk/SIMD from wall time @2.4GHz"))
for which, nm in enumerate(NAMES):
    cols = []
    for wps in (1, 2, 4, 8):
        blocks, threads = 256, 256 * wps
        ms = ctypes.c_float(); cyc = ctypes.c_double()
        ffi.check(L, L.sppark_devtest_ubench(which, 4000, blocks, threads, ctypes.byref(ms), ctypes.byref(cyc)))
        ninstr = 8 * (2 if which in (2, 3) else 1)
        if which == 12: ninstr = 8
        per_simd = (blocks * threads / 64) / 1024 * 4000 * ninstr          # wave-instructions per SIMD
        clk = ms.value * 1e-3 * 2.4e9
        cols.append("%d: %6.1f cyc, %.3f instr/clk" % (wps, cyc.value, per_simd / clk))
    print("%-28s %s" % (nm, " | ".join(cols)))
