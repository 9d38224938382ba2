"""Every order x direction x type of the single-word NTT at 2^LG on a device-resident buffer (HIP events on a non-null
stream): what the NN / RR orders and the coset transforms cost on top of NR / RN.
    NTT_FIELDS=gl64,bb31 NTT_LGS=18,24 python tools/gpu_ntt_orders.py
A tuning build (SPPARK_LIBDIR=lib_tuning) reads SPPARK_NTT_COSET_FOLD=0 (the separate scaling launch) for the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sppark_amd
from sppark_amd import ntt as N

torch.cuda.set_stream(torch.cuda.Stream())
stream = torch.cuda.current_stream().cuda_stream
FIELDS = os.environ.get("NTT_FIELDS", "gl64,bb31").split(",")
LGS = [int(v) for v in os.environ.get("NTT_LGS", "24").split(",")]
print("knobs:", {k: v for k, v in os.environ.items() if k.startswith("SPPARK_")}, flush=True)
for field in FIELDS:
    dt = torch.int64 if field.startswith("gl64") else torch.int32
    for lg in LGS:
        n = 1 << lg
        x = torch.randint(0, 2**30, (n,), dtype=dt, device="cuda")
        for typ, tname in ((0, "standard"), (1, "coset")):
            row = []
            for direction, dname in ((0, "fwd"), (1, "inv")):
                for order, oname in ((1, "NR"), (2, "RN"), (0, "NN"), (3, "RR")):
                    for _ in range(3):
                        N.compute_ntt(0, x, order, direction, typ, field, stream=stream)
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    reps = 20
                    e0.record()
                    for _ in range(reps):
                        N.compute_ntt(0, x, order, direction, typ, field, stream=stream)
                    e1.record(); torch.cuda.synchronize()
                    row.append("%s %s %.4f" % (dname, oname, e0.elapsed_time(e1) / reps))
            print("%s 2^%d %-8s ms: %s" % (field, lg, tname, " | ".join(row)), flush=True)
