"""NTT timing under explicit tile-geometry knobs: FIELD LG then triples SMAX:LGC:LGTILE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import NTTInputOutputOrder as Ord
field, lg = sys.argv[1], int(sys.argv[2])
dt = torch.int64 if field == "gl64" else torch.int32
x = torch.randint(0, 2**30, (1 << lg,), dtype=dt, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[3:]:
    smax, lgc, lgt = spec.split(":")
    os.environ["SPPARK_NTT_SMAX"] = smax; os.environ["SPPARK_NTT_LGC"] = lgc; os.environ["SPPARK_NTT_LGTILE"] = lgt
    for _ in range(3):
        sppark_amd.NTT(0, x, Ord.NR, field, stream=s)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sppark_amd.NTT(0, x, Ord.NR, field, stream=s)
    e1.record(); torch.cuda.synchronize()
    print("%s 2^%d smax=%s lgC=%s lgtile=%s: %.3f ms" % (field, lg, smax, lgc, lgt, e0.elapsed_time(e1) / 20), flush=True)
