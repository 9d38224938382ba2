"""sppark_amd's NTT and the reference's own NTT (its HIP path, oracle/_ref/libref_ntt_*.so, `make -C oracle ref_ntt`)
timed on the SAME device-resident buffers of one MI355X: events on the stream each library launches on, 20 back-to-back
transforms after a warm-up.  Also the through-the-FFI form (compute_ntt on a host buffer), which is the only one the
reference exports.  The outputs are compared in tests/test_ntt_vs_reference_gpu.py; this only times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sppark_amd
import oracle as O

torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
stream = torch.cuda.current_stream().cuda_stream
REPS = 20
MODES = (("fwd NR", 1, 0, 0), ("inv RN", 2, 1, 0), ("fwd NN", 0, 0, 0), ("coset fwd NR", 1, 0, 1))
FIELDS = os.environ.get("NTT_FIELDS", "gl64,bb31,bls12_381,bn254").split(",")
for field, dt, eb in (("gl64", torch.int64, 8), ("bb31", torch.int32, 4), ("bls12_381", torch.int64, 32), ("bn254", torch.int64, 32)):
    if field not in FIELDS or not O.ref_ntt_available(field):
        continue
    for lg in (16, 20, 22, 24) + ((26,) if eb < 32 else ()):
        n = 1 << lg
        x = torch.randint(0, 2**30, (n * (eb // 8 if eb >= 8 else 1),), dtype=dt, device="cuda")
        row = []
        for name, order, direction, typ in MODES:
            for _ in range(3):
                sppark_amd.compute_ntt(0, x, order, direction, typ, field, stream=stream)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                sppark_amd.compute_ntt(0, x, order, direction, typ, field, stream=stream)
            e1.record(); torch.cuda.synchronize()
            ours = e0.elapsed_time(e1) / REPS
            ref = O.ref_ntt_dev_ms(field, x.data_ptr(), lg, order, direction, typ, REPS)
            row.append("%s %.3f vs %.3f ms (x%.2f)" % (name, ours, ref, ref / ours))
        print("%-9s 2^%d  ours vs reference: %s" % (field, lg, " | ".join(row)), flush=True)
    # through the FFI: host buffer in, host buffer out (poc/ntt-cuda/cuda/ntt_api.cu:25-36 is exactly this)
    lg = 24 if eb < 32 else 22
    h = (np.random.default_rng(1).integers(0, 2**30, size=(1 << lg) * max(1, eb // 8), dtype=np.uint64)).astype(np.uint32 if eb == 4 else np.uint64)
    t = {}
    for who in ("ours", "reference"):
        best = 1e9
        for it in range(4):
            y = h.copy()
            t0 = time.perf_counter()
            if who == "ours":
                sppark_amd.compute_ntt(0, y, 1, 0, 0, field)
            else:
                L = O.ref_ntt_lib(field)
                err = L.compute_ntt(0, y.ctypes.data, lg, 1, 0, 0)
                assert err.code == 0
            dt_ = (time.perf_counter() - t0) * 1e3
            if it:
                best = min(best, dt_)
        t[who] = best
    print("%-9s 2^%d  compute_ntt on a host buffer (fwd NR, best of 3): ours %.2f ms, reference %.2f ms (x%.2f)"
          % (field, lg, t["ours"], t["reference"], t["reference"] / t["ours"]), flush=True)
