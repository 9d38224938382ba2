"""CPU: invariants of the MSM plan (sppark_amd/csrc/msm/msm_plan.hpp, compiled for the host by tests/emu/emu_plan.cpp)
over every size and a grid of tunables -- what the kernels' launch shapes, LDS sizes and index widths rely on."""
import ctypes
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
KEYS = ("n", "wbits", "nwins", "NB", "nbits", "HB", "LB", "NA", "L", "chunks_per_win", "nslabs", "slab_sz", "F", "K", "K1", "G", "wpg", "big", "IB", "SH", "NG")


@pytest.fixture(scope="module")
def plan_lib():
    so, src = os.path.join(EMU, "libemu_plan.so"), os.path.join(EMU, "emu_plan.cpp")
    hdr = os.path.join(os.path.dirname(HERE), "sppark_amd", "csrc", "msm", "msm_plan.hpp")
    if not os.path.exists(so) or os.stat(so).st_mtime < max(os.stat(src).st_mtime, os.stat(hdr).st_mtime):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    L = ctypes.CDLL(so)
    L.emu_make_plan.argtypes = [ctypes.c_size_t] + [ctypes.c_uint] * 9 + [ctypes.POINTER(ctypes.c_uint)]
    L.emu_make_fixed_plan.argtypes = [ctypes.c_size_t] + [ctypes.c_uint] * 3 + [ctypes.POINTER(ctypes.c_uint)]
    L.emu_make_plan_resident.argtypes = [ctypes.c_size_t, ctypes.c_uint, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint)]
    return L


def _plan(lib, n, bits=255, **kw):
    out = (ctypes.c_uint * 21)()
    lib.emu_make_plan(n, bits, kw.get("wbits", 0), kw.get("L", 0), kw.get("F", 0), kw.get("K", 0), kw.get("nslabs", 0),
                    kw.get("LB", 0), kw.get("groups", 0), kw.get("K1", 0), out)
    return dict(zip(KEYS, out))


def _check(p, n, bits):
    assert p["n"] == n and p["nbits"] == bits
    assert 2 <= p["wbits"] <= 24
    assert (p["nwins"] - 1) * p["wbits"] < bits <= p["nwins"] * p["wbits"]          # even split: the longest window
    assert p["nwins"] <= 128                                                          # msm_t::MAX_WINS
    assert p["NB"] == 1 << (p["wbits"] - 1)
    assert p["HB"] + p["LB"] == p["wbits"] - 1 and p["NA"] == 1 << p["HB"]
    assert p["LB"] <= 13 and p["HB"] <= 15                                            # LDS counters of level B / of the level-A histogram
    assert p["L"] >= 1 and p["chunks_per_win"] * p["L"] >= n > (p["chunks_per_win"] - 1) * p["L"]
    assert p["nslabs"] >= 1 and p["slab_sz"] * p["nslabs"] >= n
    # 4-byte level-A records (msm_sort_records.hpp): sign | index mod 2^IB | k_lo fills 32 bits; an index group is a whole
    # number (2^SH) of power-of-two slabs; every index is below NG groups; level B keeps at most 128 boundaries; no empty slab
    if p["IB"]:
        assert p["IB"] + p["LB"] == 31 and p["slab_sz"] & (p["slab_sz"] - 1) == 0 and p["slab_sz"] << p["SH"] == 1 << p["IB"]
        assert p["NG"] == ((p["nslabs"] - 1) >> p["SH"]) + 1 and 1 <= p["NG"] <= 128 and n <= p["NG"] << p["IB"]
        assert (p["nslabs"] - 1) * p["slab_sz"] < n and p["nslabs"] <= 129
    else:
        assert p["SH"] == 0 and p["NG"] == 1
    assert p["F"] >= 4                                                                # (< 3 would never shrink the record list)
    for k in ("K", "K1"):
        assert p[k] >= 1 and p[k] & (p[k] - 1) == 0 and p[k] <= p["NB"] and p["NB"] % p[k] == 0
    assert 1 <= p["G"] <= p["nwins"] and p["wpg"] * p["G"] >= p["nwins"] > p["wpg"] * (p["G"] - 1)


def test_automatic_plans_over_every_size(plan_lib):
    for bits in (255, 254, 253):
        for lg in range(0, 32):
            for n in {1 << lg, (1 << lg) + 1, (1 << lg) * 3 // 2 + 7, max(1, (1 << lg) - 1)}:
                if n > 1 << 31:
                    continue
                p = _plan(plan_lib, n, bits)
                _check(p, n, bits)
                # the automatic split: level A fits the LDS-staged scatter (2^12 partitions), partitions of ~2^14 entries
                assert p["NA"] <= 4096, (n, p)
                if (1 << 15) <= n <= (1 << 26):                                       # (above 2^26 points the 2^12 partitions outgrow
                    assert n // p["NA"] <= 18432, (n, p)                              # level B's register form: its two-pass form sorts them)


def test_run_length_fits_whole_rounds_of_resident_waves(plan_lib):
    """With the device's resident k_accumulate lanes R known (the driver's occupancy query: 131072 for the 14-limb fields,
    196608 for the 10-limb ones), the automatic run length makes windows x ceil(n / L) lanes fit k rounds of R exactly:
    never more rounds x run length than the power-of-two choice, every invariant intact, and the cases that prompted it."""
    for R in (65536, 131072, 196608):
        for bits in (255, 254):
            for lg in range(12, 29):
                for n in {1 << lg, (1 << lg) + 1, (1 << lg) * 3 // 2 + 7, (1 << lg) - 1, (1 << lg) * 5 // 4}:
                    out = (ctypes.c_uint * 21)()
                    plan_lib.emu_make_plan_resident(n, bits, R, out)
                    p = dict(zip(KEYS, out))
                    _check(p, n, bits)
                    q = _plan(plan_lib, n, bits)                                      # without the fit: the power-of-two run length
                    assert (p["wbits"], p["nwins"], p["NA"], p["K"], p["K1"]) == (q["wbits"], q["nwins"], q["NA"], q["K"], q["K1"])
                    groups = lambda pl: pl["nwins"] * -(-pl["chunks_per_win"] // 256)  # the launch: ceil(chunks / 256) groups of 256 lanes per window
                    rounds = lambda pl: -(-groups(pl) // (R // 256))
                    assert rounds(p) * p["L"] <= rounds(q) * q["L"], (n, R, p, q)
                    assert 4 <= p["L"] <= 1024
                    if p["L"] != q["L"]:
                        assert 2 <= rounds(q) <= 64 and rounds(p) * p["L"] * 100 <= rounds(q) * q["L"] * 92, (n, R, p, q)
                        assert groups(p) <= rounds(p) * (R // 256), (n, R, p)
                        if p["L"] > 4:                                                # ... tightly: one entry less per run would not fit
                            assert p["nwins"] * -(-(-(-n // (p["L"] - 1))) // 256) > rounds(p) * (R // 256), (n, R, p)
    out = (ctypes.c_uint * 21)()
    for n, L in ((1 << 17, 20), (1 << 18, 35), (300000, 40), (1 << 22, 128), (12000000, 129), (1 << 13, 8), (1 << 16, 16), (1 << 19, 64), (1 << 20, 128), (1 << 21, 128), (1 << 26, 256)):
        plan_lib.emu_make_plan_resident(n, 255, 131072, out)
        assert dict(zip(KEYS, out))["L"] == L, (n, dict(zip(KEYS, out)))


def test_tuned_plans(plan_lib):
    for n in (1, 300, 5000, 1 << 16, (1 << 20) + 3, 1 << 26):
        for wbits in (0, 2, 7, 13, 19, 24):
            for L in (0, 4, 64, 256):
                for K, K1 in ((0, 0), (2, 4), (8, 16), (4, 1 << 20)):
                    for LB in (0, 1, 9, 13):
                        for groups in (0, 1, 3, 200):
                            _check(_plan(plan_lib, n, 255, wbits=wbits, L=L, K=K, K1=K1, LB=LB, groups=groups, F=3, nslabs=5), n, 255)


def test_fixed_base_plans(plan_lib):
    """the one-window plan over W * n (digit, multiple) entries (msm_driver.hpp invoke_fixed)"""
    STAGE = 18 * 1024
    for bits in (255, 254):
        for lgn in range(0, 28):
            for c in range(8, 27):
                W = -(-bits // c)
                cc = -(-bits // W)                                   # the even split the driver stores
                n = (1 << lgn) + (lgn % 3)
                if W * n >= 1 << 31:
                    continue
                out = (ctypes.c_uint * 21)()
                plan_lib.emu_make_fixed_plan(n, cc, W, STAGE, out)
                p = dict(zip(KEYS, out))
                assert p["n"] == W * n and p["nwins"] == 1 and p["wbits"] == cc and p["NB"] == 1 << (cc - 1)
                assert p["HB"] + p["LB"] == cc - 1 and p["LB"] <= 13 and p["NA"] == 1 << p["HB"]
                assert p["NA"] <= 4096 or cc - 1 - 13 > 12, (n, c, p)      # staged level A unless the window leaves no choice
                assert p["chunks_per_win"] * p["L"] >= p["n"] and p["slab_sz"] * p["nslabs"] >= p["n"]
                assert p["big"] == STAGE and p["G"] == 1 and p["K1"] <= p["NB"] and p["K"] <= p["NB"]
