import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sppark_amd
from sppark_amd import NTTInputOutputOrder as Ord
lg = int(sys.argv[1])
x = torch.randint(0, 2**62, ((1 << lg) * 4,), dtype=torch.int64, device="cuda"); x[3::4] &= 0x0fffffffffffffff
torch.cuda.set_stream(torch.cuda.Stream())                  # non-null: on the NULL stream sppark_ntt synchronises after every call
s = torch.cuda.current_stream().cuda_stream
for spec in sys.argv[2:]:
    smax, lgc, lgt = spec.split(":")
    os.environ["SPPARK_NTT_SMAX"] = smax; os.environ["SPPARK_NTT_LGC"] = lgc; os.environ["SPPARK_NTT_LGTILE"] = lgt
    for _ in range(2):
        sppark_amd.NTT(0, x, Ord.NR, "bls12_381", stream=s)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        sppark_amd.NTT(0, x, Ord.NR, "bls12_381", stream=s)
    e1.record(); torch.cuda.synchronize()
    print("bls12_381 Fr 2^%d smax=%s lgC=%s lgtile=%s: %.3f ms" % (lg, smax, lgc, lgt, e0.elapsed_time(e1) / 5), flush=True)
