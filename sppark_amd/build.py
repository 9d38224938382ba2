"""Build the sppark_amd C-ABI libraries for gfx950 with hipcc (cross-compiles
without a GPU).  One shared object per FEATURE, as the reference builds one per
-DFEATURE_* (rust/src/build.rs ccmd(); poc/*/build.rs).  Translation units are
compiled in parallel; a library is re-linked when the contents of csrc/ change, a translation
unit is recompiled only when one of the files it includes (hipcc -MD) or its flags changed.

    python -m sppark_amd.build [--force] [--only bls12_381,gl64]
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SPPARK_LIBDIR=<name> (a directory beside lib/, e.g. lib_tuning): a second set of libraries with other flags for an A/B
# job -- "SPPARK_LIBDIR=lib_tuning SPPARK_EXTRA_FLAGS=-DSPPARK_TUNING python -m sppark_amd.build --only gl64,bb31" -- which
# the tools select with the same variable (sppark_amd/ffi.py).  The tests and bench.py never set it.
_LIBNAME = os.environ.get("SPPARK_LIBDIR", "lib")
LIBDIR = os.path.join(HERE, _LIBNAME)
OBJDIR = os.path.join(os.path.dirname(HERE), "build", "obj" if _LIBNAME == "lib" else "obj_" + _LIBNAME)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wno-duplicate-decl-specifier"]
FLAGS += os.environ.get("SPPARK_EXTRA_FLAGS", "").split()      # experiments: extra -D switches for every translation unit

MSM_TUS = ["api/msm_api.hip", "msm/k_accumulate.hip", "msm/k_reduce.hip",
           "msm/k_bucket1.hip", "msm/k_bucketN.hip", "msm/k_bucket_lat.hip",
           "msm/k_accumulate.hip:SPPARK_G2", "msm/k_reduce.hip:SPPARK_G2",       # the same kernels over Fp2 (G2)
           "msm/k_bucket1.hip:SPPARK_G2", "msm/k_bucketN.hip:SPPARK_G2",
           "api/ntt_api.hip:SPPARK_NTT_WITH_MSM",          # compute_ntt over the curve's scalar field
           "ntt/k_ntt_pass.hip:SPPARK_NTT_DIF=1", "ntt/k_ntt_pass.hip:SPPARK_NTT_DIF=0"]
MSM_G1_TUS = [t for t in MSM_TUS if "SPPARK_G2" not in t]      # curves without a G2 (Pasta)
NTT_TUS = ["api/ntt_api.hip", "ntt/k_ntt_pass.hip:SPPARK_NTT_DIF=1", "ntt/k_ntt_pass.hip:SPPARK_NTT_DIF=0",
           "ntt/k_ntt_r64.hip:SPPARK_NTT_DIF=1", "ntt/k_ntt_r64.hip:SPPARK_NTT_DIF=0"]       # single-word fields: radix-64 plan

TARGETS = {
    "bls12_381": ("FEATURE_BLS12_381", MSM_TUS),
    "bn254":     ("FEATURE_BN254", MSM_TUS),
    "bls12_377": ("FEATURE_BLS12_377", MSM_TUS),
    "pallas":    ("FEATURE_PALLAS", MSM_G1_TUS),
    "vesta":     ("FEATURE_VESTA", MSM_G1_TUS),
    "gl64":      ("FEATURE_GOLDILOCKS", NTT_TUS),
    "bb31":      ("FEATURE_BABY_BEAR", NTT_TUS),
    # field types without NTT parameters in the reference: polynomial primitives only
    "m31":       ("FEATURE_MERSENNE31", ["api/poly_only_api.hip"]),
    "bb31x4":    ("FEATURE_BABY_BEAR_X4", ["api/poly_only_api.hip"]),
    # the reference's compile-time root conventions (ntt/parameters/goldilocks.h:7-82, baby_bear.h:7-74)
    "gl64_plonky2":   ("FEATURE_GOLDILOCKS -DGOLDILOCKS_PLONKY2", NTT_TUS),
    "bb31_canonical": ("FEATURE_BABY_BEAR -DBABY_BEAR_CANONICAL", NTT_TUS),
    # test-only libraries (device test hooks + micro-benchmarks): never linked into the product ones
    "bls12_381_devtest": ("FEATURE_BLS12_381", ["api/devtest_api.hip"]),
    "bn254_devtest":     ("FEATURE_BN254", ["api/devtest_api.hip"]),
    "bls12_377_devtest": ("FEATURE_BLS12_377", ["api/devtest_api.hip"]),
    "pallas_devtest":    ("FEATURE_PALLAS", ["api/devtest_api.hip"]),
    "vesta_devtest":     ("FEATURE_VESTA", ["api/devtest_api.hip"]),
    "gl64_devtest":      ("FEATURE_GOLDILOCKS", ["api/devtest_small_api.hip"]),
    "bb31_devtest":      ("FEATURE_BABY_BEAR", ["api/devtest_small_api.hip"]),
}
PRODUCT = ("bls12_381", "bn254", "bls12_377", "pallas", "vesta", "gl64", "bb31", "gl64_plonky2", "bb31_canonical", "m31", "bb31x4")


def lib_path(name):
    return os.path.join(LIBDIR, "libsppark_%s.so" % name)


def _sources_stamp():
    h = hashlib.sha1()
    for root, _, files in sorted(os.walk(CSRC)):
        for f in sorted(files):
            p = os.path.join(root, f)
            h.update(os.path.relpath(p, CSRC).encode()); h.update(open(p, 'rb').read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _deps_changed(obj, cmdline):
    """True when |obj| must be recompiled: missing, compiled with other flags, or older than one of
    the files its depfile (-MD) lists.  Keeps an edit of one kernel header from rebuilding the lot."""
    dep, flg = obj + ".d", obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(flg)):
        return True
    if open(flg).read() != cmdline:
        return True
    t = os.path.getmtime(obj)
    txt = open(dep).read().replace("\\\n", " ")
    for tok in txt.split()[1:]:
        if tok.endswith(":"):
            continue
        try:
            if os.path.getmtime(tok) > t:
                return True
        except OSError:
            return True
    return False


def _compile(job):
    src, obj, feature = job
    t0 = time.time()
    src, _, extra = src.partition(":")
    cmd = [HIPCC] + FLAGS + ("-D" + feature).split() + (["-D" + extra] if extra else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    cmdline = " ".join(cmd)
    if not _deps_changed(obj, cmdline):
        return src, feature, 0.0, 0, "", False
    r = subprocess.run(cmd + ["-MD", "-MF", obj + ".d"], capture_output=True, text=True)
    if r.returncode == 0:
        open(obj + ".cmd", "w").write(cmdline)
    return src, feature, time.time() - t0, r.returncode, r.stderr, True


def build(only=None, force=False, verbose=True, jobs=None):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = _sources_stamp()
    names = [n for n in TARGETS if (only is None or n in only)]
    names = [n for n in names if all(os.path.exists(os.path.join(CSRC, s.split(":")[0])) for s in TARGETS[n][1])]
    todo, links = [], {}
    for n in names:
        feature, tus = TARGETS[n]
        stamp_file = lib_path(n) + ".stamp"
        if not force and os.path.exists(lib_path(n)) and os.path.exists(stamp_file) \
                and open(stamp_file).read() == stamp:
            continue
        objs = []
        for s in tus:
            obj = os.path.join(OBJDIR, "%s__%s.o" % (n.replace("_devtest", ""), s.replace(":", "__").replace("/", "_")))
            objs.append(obj)
            if force and os.path.exists(obj + ".cmd"):
                os.remove(obj + ".cmd")
            todo.append((s, obj, feature))
        links[n] = objs
    # longest translation units first (measured seconds, BLS12-381 / alt_bn128 roughly 2:1):
    # with 8 cores the makespan is then bounded by total work, not by a late long job
    cost = {"msm/k_bucketN.hip": 85, "msm/k_bucket_lat.hip": 95, "msm/k_bucketN.hip:SPPARK_G2": 75, "msm/k_accumulate.hip:SPPARK_G2": 70,
            "ntt/k_ntt_pass.hip:SPPARK_NTT_DIF=0": 68, "ntt/k_ntt_pass.hip:SPPARK_NTT_DIF=1": 57,
            "msm/k_reduce.hip:SPPARK_G2": 54, "msm/k_bucket1.hip:SPPARK_G2": 53, "api/devtest_api.hip": 100,
            "msm/k_bucket1.hip": 35, "api/msm_api.hip": 28, "msm/k_accumulate.hip": 25, "msm/k_reduce.hip": 24}
    todo.sort(key=lambda job: -cost.get(job[0], 5) * (2 if "BLS12" in job[2] else 1))
    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or os.cpu_count() or 4) as ex:
            for src, feature, dt, rc, err, ran in ex.map(_compile, todo):
                if verbose and ran:
                    print("[sppark_amd.build] %-24s %-20s %6.1fs" % (src, feature, dt), flush=True)
                if rc != 0:
                    raise RuntimeError("hipcc failed for %s (%s):\n%s" % (src, feature, err))
    for n, objs in links.items():
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(n)] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed for %s:\n%s" % (n, r.stderr))
        open(lib_path(n) + ".stamp", "w").write(stamp)
        if verbose:
            print("[sppark_amd.build] linked", lib_path(n), flush=True)
    return [lib_path(n) for n in names]


if __name__ == "__main__":
    only = None
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1].split(",")
    build(only=only, force="--force" in sys.argv)
