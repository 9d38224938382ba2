"""CPU: register / scratch budgets of the hot kernels, read from the objects the build left under build/obj
(tools/isa_stats.py: kernel-descriptor metadata of the gfx950 code object; no GPU needed).

Why this is a test: in round 3 a harmless-looking change of a zero test inside the mixed addition made hipcc
allocate 269 registers for k_accumulate instead of 238 -- ONE wave per SIMD instead of two, 151 ms instead of
114 ms for the 2^26-point accumulation -- and nothing but a GPU timing showed it.  Occupancy steps on gfx950:
<= 256 registers per lane (VGPR + AGPR) for two waves per SIMD, <= 168 for three, <= 128 for four, <= 64 for eight."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OBJ = os.path.join(ROOT, "build", "obj")


def _kernels(obj):
    import isa_stats
    path = os.path.join(OBJ, obj)
    if not os.path.exists(path):
        pytest.skip("%s not built here" % obj)
    meta = isa_stats.metadata(isa_stats.code_object(path))
    assert meta, obj
    return meta


def _regs(md):
    return int(md.get("vgpr_count") or 0) + int(md.get("agpr_count") or 0)


@pytest.mark.parametrize("obj,needle,max_regs", [
    ("bls12_381__msm_k_accumulate.hip.o", "k_accumulate", 256),          # two waves per SIMD
    ("bls12_377__msm_k_accumulate.hip.o", "k_accumulate", 256),
    ("bn254__msm_k_accumulate.hip.o", "k_accumulate", 168),              # ten-limb fields: three waves
    ("bls12_381__msm_k_bucket1.hip.o", "k_bucket_level1", 256),
    ("bls12_381__msm_k_reduce.hip.o", "k_join_runs", 256),
    ("gl64__ntt_k_ntt_r64.hip__SPPARK_NTT_DIF=1.o", "k_ntt", 64),        # eight waves per SIMD
    ("gl64__ntt_k_ntt_r64.hip__SPPARK_NTT_DIF=0.o", "k_ntt", 64),
])
def test_register_budget(obj, needle, max_regs):
    meta = _kernels(obj)
    hit = {k: v for k, v in meta.items() if needle in k and "vgpr_count" in v}
    assert hit, (obj, needle)
    for name, md in hit.items():
        assert _regs(md) <= max_regs, (name[:60], md)
        if "k_accumulate" in needle and max_regs == 256:
            assert int(md.get("private_segment_fixed_size") or 0) == 0, (name[:60], "scratch", md)


@pytest.mark.parametrize("obj", ["bls12_381__msm_k_reduce.hip.o", "bls12_381__msm_k_reduce.hip__SPPARK_G2.o",
                                 "bls12_381__msm_k_bucketN.hip.o", "bls12_381__msm_k_bucketN.hip__SPPARK_G2.o",
                                 "bls12_381__msm_k_bucket1.hip.o", "bls12_381__msm_k_accumulate.hip.o",
                                 "bls12_381__msm_k_bucket_lat.hip.o", "bn254__msm_k_bucket_lat.hip.o"])      # incl. the cooperative kernels
def test_point_arithmetic_kernels_are_one_wave_per_simd_groups(obj):
    """The kernels that do point arithmetic are built for work-groups of <= 256 lanes (one wave per SIMD, up to 512
    registers).  A larger `__launch_bounds__` caps the kernel's registers (1024 lanes: 128) -- the 14-limb addition then
    spills 187 of them, and over Fp2 the OUTLINED addition, compiled for up to 512 registers, is called from a kernel
    that owns 128: round 3's first k_reduce_tail did exactly that and hung the G2 tests on the GPU box."""
    meta = _kernels(obj)
    for name, md in meta.items():
        if "max_flat_workgroup_size" in md:
            assert int(md["max_flat_workgroup_size"]) <= 256, (name[:70], md["max_flat_workgroup_size"])


@pytest.mark.parametrize("obj,max_regs,max_scratch", [
    ("bn254__msm_k_accumulate.hip__SPPARK_G2.o", 256, 256),          # Fp2 over ten limbs: two waves per SIMD, spills bounded
    ("bls12_381__msm_k_accumulate.hip__SPPARK_G2.o", 512, 64),       # Fp2 over fourteen limbs: one wave, (almost) no scratch
    ("bls12_377__msm_k_accumulate.hip__SPPARK_G2.o", 512, 64),
])
def test_g2_accumulate_scratch_stays_resident(obj, max_regs, max_scratch):
    """The G2 accumulation over the 28-bit-limb field (ec/xyzzx2_dev.hpp).  Capping the fourteen-limb form at 256
    registers spills 852 bytes per lane: times every resident lane that is above the runtime's resident scratch limit,
    scratch is then set up per dispatch and small MSMs pay milliseconds each (measured: the G2 suite 8 s -> 123 s,
    profiles/r03_msm_g2_montx.log).  The ten-limb form's 184 bytes stay resident."""
    meta = _kernels(obj)
    hit = {k: v for k, v in meta.items() if "k_accumulate" in k and "vgpr_count" in v}
    assert hit, obj
    for name, md in hit.items():
        # (.vgpr_count of a kernel that uses accumulation registers is already the unified total)
        assert int(md.get("vgpr_count") or 0) <= max_regs, (name[:60], md)
        assert int(md.get("private_segment_fixed_size") or 0) <= max_scratch, (name[:60], "scratch", md)


def test_cooperative_kernels_fit_one_work_group_per_cu():
    """The cooperative kernels (msm_coop_kernels.hpp) are work-groups of exactly four waves, one per SIMD: <= 512 registers
    per lane, and LDS (static + the dynamic image of the top) within the CU's 160 KB.  Their scratch is the frame of the
    out-of-line serial addition of the exceptional lanes only -- when that call took the operands by reference they lived
    in scratch on the hot path and every kernel was 25 % slower (profiles/r04_msm_coop_ab.log): bounded here."""
    meta = _kernels("bls12_381__msm_k_bucket_lat.hip.o")
    hit = {k: v for k, v in meta.items() if "coop" in k and "vgpr_count" in v}
    assert len(hit) >= 6, sorted(hit)
    for name, md in hit.items():
        assert int(md.get("max_flat_workgroup_size") or 0) == 256, (name[:60], md)
        # (.vgpr_count of a kernel that uses accumulation registers is already the unified total)
        assert int(md.get("vgpr_count") or 0) <= 512, (name[:60], md)
        # static LDS: the exchange area (+ the image of a tree / a level); the top's image of 2^nl points is dynamic and
        # sized by the driver (top_bits_coop_lds) under the 160 KB of a CU
        # (k_bucket_small_bits_coop holds the 128 buckets of a subset sum beside the exchange area: 56 KB, two work-groups per CU)
        assert int(md.get("group_segment_fixed_size") or 0) <= (60 if "small_bits" in name else 40) * 1024, (name[:60], md)
        assert int(md.get("private_segment_fixed_size") or 0) <= 1280, (name[:60], "scratch", md)


@pytest.mark.parametrize("obj", ["bls12_381__msm_k_accumulate.hip__SPPARK_G2.o", "bn254__msm_k_accumulate.hip__SPPARK_G2.o",
                                 "bls12_377__msm_k_accumulate.hip__SPPARK_G2.o"])
def test_g2_one_component_per_wave_kernel_fits_two_waves_per_simd(obj):
    """k_accumulate_g2c (msm_g2c_kernels.hpp; the default G2 accumulation over the 14-limb fields): a pair of waves per 64 chunks is only worth having
    if BOTH fit a SIMD twice over -- at most 256 registers, no scratch (the two earlier attempts at two G2 waves per
    SIMD died of spills), and four 128-lane work-groups' exchange areas within a CU's 160 KB of LDS."""
    meta = _kernels(obj)
    hit = {k: v for k, v in meta.items() if "k_accumulate_g2c" in k and "vgpr_count" in v}
    assert len(hit) == 1, sorted(meta)
    md = next(iter(hit.values()))
    assert int(md.get("max_flat_workgroup_size") or 0) == 128
    assert int(md.get("vgpr_count") or 0) <= 256, md
    assert int(md.get("private_segment_fixed_size") or 0) == 0, md
    assert 4 * int(md.get("group_segment_fixed_size") or 0) <= 160 * 1024, md


def test_g2_wave_pair_kernel_keeps_its_barriers_paired():
    """k_accumulate_g2c branches on the wave's role into two separately inlined walks, each with its own copies of the
    work-group barriers (msm_g2c_kernels.hpp): correct because s_barrier on gfx9 counts WAVES, not program addresses, and both
    walks execute the same number of barriers in the same order (the host emulation runs the pair as threads over a counting
    barrier and would hang otherwise).  What a compiler could do to that -- merge the tails of the two walks, drop or
    duplicate a barrier on one side -- shows in the emitted code: the static number of s_barrier instructions of the
    kernel must stay even, half of them before the second walk's first barrier (the walks are emitted one after the
    other), and the object must have been built for gfx950 (the one architecture this reasoning is made for)."""
    import subprocess
    import isa_stats
    path = os.path.join(OBJ, "bls12_381__msm_k_accumulate.hip__SPPARK_G2.o")
    if not os.path.exists(path):
        pytest.skip("not built here")
    co = isa_stats.code_object(path)
    hdr = subprocess.run([isa_stats.LLVM + "/llvm-readelf", "-h", co], capture_output=True, text=True).stdout
    assert "gfx950" in hdr, hdr[-400:]
    dis = subprocess.run([isa_stats.LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    body, on = [], False
    for line in dis.splitlines():
        if line.endswith(">:"):
            on = "k_accumulate_g2c" in line
            continue
        if on:
            body.append(line.split()[0] if line.split() else "")
    bars = [i for i, op in enumerate(body) if op == "s_barrier"]
    assert len(bars) >= 8 and len(bars) % 2 == 0, len(bars)
    # the same number of instructions between consecutive barriers in both walks, within the difference the two components'
    # arithmetic makes (component 1 of a product is a sum, component 0 a difference of two base products)
    half = len(bars) // 2
    gaps0 = [bars[i + 1] - bars[i] for i in range(half - 1)]
    gaps1 = [bars[half + i + 1] - bars[half + i] for i in range(half - 1)]
    for a, b in zip(gaps0, gaps1):
        assert abs(a - b) <= 0.25 * max(a, b) + 192, (gaps0, gaps1)


@pytest.mark.parametrize("unit,feature", [("api/ntt_api.hip", "FEATURE_GOLDILOCKS"), ("api/ntt_api.hip", "FEATURE_BLS12_381 -DSPPARK_NTT_WITH_MSM"),
                                          ("api/msm_api.hip", "FEATURE_BN254")])
def test_tuning_build_still_compiles(unit, feature):
    """The experiment knobs live behind -DSPPARK_TUNING (a second set of libraries for A/B jobs, sppark_amd/build.py
    SPPARK_LIBDIR); nothing in the default build or its tests compiles that code, so it is syntax-checked here -- a tuning
    build that silently failed once left a job measuring a stale library."""
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "sppark_amd", "csrc", unit)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-Wno-duplicate-decl-specifier", "-DSPPARK_TUNING"]
                       + ("-D" + feature).split() + [src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_host_field_adc_chains_match_plain_cios(tmp_path):
    """ff/mont_host.hpp (the Horner over the window sums every MSM ends in): the adc-chain product and the dedicated square
    against the plain CIOS loop on random and edge operands (all-ones limbs, p - 1), three moduli of 6 / 4 / 6 limbs.  Built with
    the product's own host compiler (hipcc's clang: the builtins the fast path needs)."""
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("clang++ of the ROCm toolchain not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the product as built (mulx / adcx / adox where the host has them: ff/mont_host_x86.hpp) and the adc-chain C product alone
    for extra in ([], ["-DSPPARK_HOST_NO_MULX"]):
        exe = str(tmp_path / ("host_field_bench" + str(len(extra))))
        subprocess.check_call([clang, "-O2", "-std=c++17", "-I", os.path.join(root, "sppark_amd", "csrc")] + extra +
                              [os.path.join(root, "tools", "host_field_bench.cpp"), "-o", exe])
        out = subprocess.run([exe, "20000"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count("mismatches 0;") == 3, out.stdout
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_mont_host_x86.py")], capture_output=True, text=True)
    assert gen.stdout == open(os.path.join(root, "sppark_amd", "csrc", "ff", "mont_host_x86.hpp")).read()      # the header is the generator's output
    src = open(os.path.join(root, "sppark_amd", "csrc", "ff", "mont_host.hpp")).read()
    assert "SPPARK_HOST_ADC" in src
