"""Small transforms (2^8 .. 2^20), ours against the reference's HIP build, device-resident: both timed by events around
REPS back-to-back calls issued from C loops on a non-null stream (ours: tools/ntt_loop.cpp calling sppark_ntt; the
reference: ref_ntt_dev_timed of oracle/ref_ntt_shim.cu).  Next to it the latency a compute_ntt-style caller sees on the
NULL stream, where both libraries return only when the result is in place (ours: sppark_ntt(stream = NULL); the
reference: ref_ntt_dev, which synchronises): wall clock per call from a Python loop, the same loop for both.

    python tools/gpu_ntt_small_vs_reference.py [field=gl64 | fields=gl64,bb31] [lg=8 | lgs=8-11] [order=1] [only=ours|ref]
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle as O
from sppark_amd import ffi

REPS = 400
args = dict(a.split("=") for a in sys.argv[1:])
_so = os.path.join(ROOT, "tools", "libntt_loop.so")
if not os.path.exists(_so):                                  # (host code only; built once, travels with the snapshot)
    import subprocess
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-fPIC", "-shared", os.path.join(ROOT, "tools", "ntt_loop.cpp"), "-o", _so])
loop = ctypes.CDLL(_so)
loop.ntt_loop.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
torch.cuda.set_stream(torch.cuda.Stream())
stream = torch.cuda.current_stream().cuda_stream
order = int(args.get("order", 1))
behind = []
# warm-up: the first ~10 ms of synchronous calls of a process run at about half speed on these boxes (whichever library makes
# them: the first row of every earlier log shows 27 us where every later row shows 14); 1500 untimed calls of either side first
_w = torch.zeros(256, dtype=torch.int64, device="cuda")
for _ in range(1500):
    ffi.load("gl64").sppark_ntt(0, ctypes.c_void_p(_w.data_ptr()), 8, 1, 0, 0, None)
if args.get("only", "ref") == "ref":
    for _ in range(1500):
        O.ref_ntt_lib("gl64").ref_ntt_dev(ctypes.c_void_p(_w.data_ptr()), 8, 1, 0, 0)
torch.cuda.synchronize()
for field, dt, eb in (("gl64", torch.int64, 8), ("bb31", torch.int32, 4), ("bls12_381", torch.int64, 32), ("bn254", torch.int64, 32)):
    if field not in args.get("fields", args.get("field", field)).split(","):
        continue
    L = ffi.load(field)
    fn = ctypes.cast(L.sppark_ntt, ctypes.c_void_p)
    lo_hi = [int(v) for v in args.get("lgs", "8-20").split("-")]
    for lg in ([int(args["lg"])] if "lg" in args else range(lo_hi[0], lo_hi[1] + 1)):
        n = 1 << lg
        x = torch.randint(0, 2**30, (n * (eb // 8 if eb >= 8 else 1),), dtype=dt, device="cuda")
        p = ctypes.c_void_p(x.data_ptr())
        out, ours, ref = [], None, None
        if args.get("only", "ours") == "ours":
            ms, iss = ctypes.c_float(0), ctypes.c_float(0)
            rc = loop.ntt_loop(fn, p, lg, order, 0, 0, stream, REPS, ctypes.byref(ms), ctypes.byref(iss))
            assert rc == 0, rc
            ours = ms.value
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(REPS):
                L.sppark_ntt(0, p, lg, order, 0, 0, None)
            null_ours = (time.perf_counter() - t) / REPS * 1e3
            out.append("ours %.4f ms (issue %.1f us/call; NULL stream, synchronous: %.4f ms)" % (ours, iss.value, null_ours))
        if args.get("only", "ref") == "ref":
            ref = O.ref_ntt_dev_ms(field, x.data_ptr(), lg, order, 0, 0, REPS)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(REPS):
                O.ref_ntt_lib(field).ref_ntt_dev(p, lg, order, 0, 0)
            null_ref = (time.perf_counter() - t) / REPS * 1e3
            out.append("reference %.4f ms (synchronous: %.4f ms)" % (ref, null_ref))
        if ours is not None and ref is not None and ours > ref:
            behind.append((field, lg, ours, ref))
        print("%-9s 2^%-2d order %d fwd: %s" % (field, lg, order, ", ".join(out)), flush=True)
print("rows where the reference's build leads: %s" % (behind if behind else "none"), flush=True)
