"""NTT timings on device-resident buffers (HIP events via torch on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sppark_amd
from sppark_amd import NTTInputOutputOrder as Ord

torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
stream = torch.cuda.current_stream().cuda_stream
FIELDS = os.environ.get("NTT_FIELDS", "gl64,bb31,bls12_381,bn254").split(",")      # e.g. NTT_FIELDS=gl64 NTT_LGS=12,16,24
LGS = [int(v) for v in os.environ["NTT_LGS"].split(",")] if os.environ.get("NTT_LGS") else None
print("plan knobs:", {k: v for k, v in os.environ.items() if k.startswith("SPPARK_NTT")}, flush=True)
for field, dt, eb in (("gl64", torch.int64, 8), ("bb31", torch.int32, 4), ("bls12_381", torch.int64, 32), ("bn254", torch.int64, 32)):
    if field not in FIELDS:
        continue
    for lg in LGS or ((16, 20, 22, 24) + ((26,) if eb < 32 else ())):
        n = 1 << lg
        x = torch.randint(0, 2**30, (n * (eb // 8 if eb >= 8 else 1),), dtype=dt, device="cuda")
        res = []
        for name, fn, order in (("fwd NR", sppark_amd.NTT, Ord.NR), ("inv RN", sppark_amd.iNTT, Ord.RN),
                                ("fwd NN", sppark_amd.NTT, Ord.NN), ("coset fwd NR", sppark_amd.coset_NTT, Ord.NR)):
            for _ in range(3):
                fn(0, x, order, field, stream=stream)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                fn(0, x, order, field, stream=stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res.append("%s %.3f ms (%.2e el/s, %.0f GB/s alg)" % (name, ms, n / ms * 1e3, 2 * eb * n / ms / 1e6))
        print("%s 2^%d: %s" % (field, lg, " | ".join(res)), flush=True)
